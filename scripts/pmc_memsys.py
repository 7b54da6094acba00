"""Memory-system counters per dispatch of one render (diagnostic, one-chain frame so that a dispatch owns the chip):
    python scripts/pmc_memsys.py <outdir> [lib.so] [spp] [scene]
Separate --pmc passes (kernel-trace only) over scripts/gpu_depth_probe.py; prints the first dispatches of the LAST render and
per-kernel totals: L2<->fabric request mix and mean outstanding level (latency = level / requests, in L2 cycles), L1<->L2 request
latency, the address-translation (UTCL1) miss rate and the stall counters of the path."""
import collections
import csv
import glob
import os
import re
import subprocess
import sys

out = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else ""
spp = sys.argv[3] if len(sys.argv) > 3 else "128"
scene = sys.argv[4] if len(sys.argv) > 4 else "sphere_scene"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TMPDIR", "/tmp")
if lib:
    os.environ["RTOW_GPU_LIB"] = os.path.abspath(lib)
GROUPS = {
    "tcc1": ["TCC_EA0_RDREQ", "TCC_EA0_RDREQ_32B", "TCC_EA0_RDREQ_64B", "TCC_EA0_RDREQ_128B"],
    "tcc2": ["TCC_EA0_RDREQ_LEVEL", "TCC_EA0_WRREQ_LEVEL", "TCC_EA0_WRREQ", "TCC_EA0_WRREQ_64B"],
    "tcc3": ["TCC_HIT", "TCC_MISS", "TCC_REQ", "TCC_TAG_STALL"],
    "tcc4": ["TCC_EA0_WRREQ_STALL", "TCC_TOO_MANY_EA_WRREQS_STALL", "TCC_EA0_RDREQ_DRAM_CREDIT_STALL", "TCC_BUSY"],
    "tcc5": ["TCC_EA0_RDREQ_DRAM", "TCC_EA0_WRREQ_DRAM", "TCC_NORMAL_WRITEBACK", "TCC_NORMAL_EVICT"],
    "tcp1": ["TCP_TCC_READ_REQ_LATENCY", "TCP_TCC_READ_REQ", "TCP_TCC_WRITE_REQ_LATENCY", "TCP_TCC_WRITE_REQ"],
    "tcp2": ["TCP_UTCL1_REQUEST", "TCP_UTCL1_TRANSLATION_MISS", "TCP_UTCL1_TRANSLATION_HIT", "TCP_UTCL1_STALL_INFLIGHT_MAX"],
    "tcp3": ["TCP_PENDING_STALL_CYCLES", "TCP_TCR_TCP_STALL_CYCLES", "TCP_GATE_EN1", "TCP_TOTAL_ACCESSES"],
    "tcp4": ["TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS", "TCP_UTCL1_SERIALIZATION_STALL", "TCP_UTCL1_THRASHING_STALL", "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS"],
    # (a pass over the TA counters TA_ADDR_STALLED_BY_TC_CYCLES, TA_DATA_STALLED_BY_TC_CYCLES, TA_TA_BUSY, TA_TOTAL_WAVEFRONTS never
    # returned on this pool, round 5: killed after 7 silent minutes; left out)
    "grbm": ["GRBM_GUI_ACTIVE", "GRBM_UTCL2_BUSY"],
    # the wave's side of the vector-memory path: SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM = mean cycles a vector-memory instruction is outstanding
    "sqmem": ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INST_LEVEL_VMEM", "SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_VMEM_TA_CMD_FIFO_FULL",
              "SQ_VMEM_WR_TA_DATA_FIFO_FULL", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR"],
    "sqmix": ["SQ_INSTS_VALU", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CU_CYCLES",
              "SQ_CYCLES", "SQ_WAVE_CYCLES"],
}
only = os.environ.get("RTOW_PMC_GROUPS")
if only:
    GROUPS = {k: v for k, v in GROUPS.items() if k in only.split(",")}
os.makedirs(out, exist_ok=True)


def short(n):
    m = re.match(r"void rt::(k_\w+)<([^>]*)>", n)
    return (m.group(1) + "<" + m.group(2).replace(" ", "") + ">") if m else n.split("(")[0].replace("rt::", "")


data = {}
for g, ctrs in GROUPS.items():
    d = os.path.join(out, g)
    if os.environ.get("RTOW_PMC_OFFLINE") != "1":  # (=1: parse the passes a previous run left in <outdir>)
        subprocess.run(["rm", "-rf", d])
        cmd = ["rocprofv3", "--pmc"] + ctrs + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.join(root, "scripts", "gpu_depth_probe.py"), spp, "0", scene]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=root)
        open(os.path.join(out, g + ".log"), "w").write(r.stdout + r.stderr)
    trace = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not trace or not cc:
        print("no output for group", g, "(see", os.path.join(out, g + ".log") + ")", flush=True)
        continue
    rows = sorted(csv.DictReader(open(trace[0])), key=lambda r: int(r["Start_Timestamp"]))
    disp = [(r["Dispatch_Id"], short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
    last = max(i for i, x in enumerate(disp) if x[1] == "k_init_counts")
    disp = disp[last:]
    vals = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc[0])):
        vals[r["Dispatch_Id"]][r["Counter_Name"]] = vals[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    data[g] = [(name, us, vals.get(did, {})) for did, name, us in disp]
    print(f"group {g}: {len(disp)} dispatches", flush=True)

if not data:
    sys.exit(1)
n = min(len(v) for v in data.values())
ref = next(iter(data.values()))
for i in range(min(n, 12)):
    name, us = ref[i][0], ref[i][1]
    merged = {}
    for g in data:
        merged.update(data[g][i][2])
    print(f"# {i:2d} {name[:36]:36s} {us:9.1f} us")
    for k in sorted(merged):
        print(f"      {k:44s} {merged[k]:18.0f}  per us {merged[k] / max(us, 1e-9):12.1f}")
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for g in data:
    for name, us, v in data[g][:n]:
        tot[name]["us:" + g] += us
        for k, x in v.items():
            tot[name][k] += x
print()
for name, v in tot.items():
    if not name.startswith("k_shade") and not name.startswith("k_intersect"):
        continue
    print(f"== {name}")
    g = lambda k: v.get(k, 0.0)
    rd = max(g("TCC_EA0_RDREQ"), 1.0)
    print(f"   L2->fabric reads {rd:.3e}: 32 B {g('TCC_EA0_RDREQ_32B') / rd:.2f}  64 B {g('TCC_EA0_RDREQ_64B') / rd:.2f}  128 B {g('TCC_EA0_RDREQ_128B') / rd:.2f}; "
          f"to DRAM {g('TCC_EA0_RDREQ_DRAM') / rd:.2f}")
    print(f"   mean read latency L2->fabric {g('TCC_EA0_RDREQ_LEVEL') / rd:.0f} L2 cycles; writes {g('TCC_EA0_WRREQ'):.3e} (64 B {g('TCC_EA0_WRREQ_64B') / max(g('TCC_EA0_WRREQ'), 1):.2f}), "
          f"mean write latency {g('TCC_EA0_WRREQ_LEVEL') / max(g('TCC_EA0_WRREQ'), 1):.0f}")
    print(f"   L2 hit rate {g('TCC_HIT') / max(g('TCC_HIT') + g('TCC_MISS'), 1):.3f} of {g('TCC_REQ'):.3e} requests; tag stall {g('TCC_TAG_STALL'):.3e}; "
          f"write-request stall {g('TCC_EA0_WRREQ_STALL'):.3e}; too many write requests {g('TCC_TOO_MANY_EA_WRREQS_STALL'):.3e}; "
          f"DRAM read credit stall {g('TCC_EA0_RDREQ_DRAM_CREDIT_STALL'):.3e}; busy {g('TCC_BUSY'):.3e}")
    print(f"   L1->L2 read latency {g('TCP_TCC_READ_REQ_LATENCY') / max(g('TCP_TCC_READ_REQ'), 1):.0f} cycles over {g('TCP_TCC_READ_REQ'):.3e} requests; "
          f"write latency {g('TCP_TCC_WRITE_REQ_LATENCY') / max(g('TCP_TCC_WRITE_REQ'), 1):.0f} over {g('TCP_TCC_WRITE_REQ'):.3e}")
    print(f"   UTCL1: {g('TCP_UTCL1_REQUEST'):.3e} requests, miss rate {g('TCP_UTCL1_TRANSLATION_MISS') / max(g('TCP_UTCL1_TRANSLATION_MISS') + g('TCP_UTCL1_TRANSLATION_HIT'), 1):.4f}, "
          f"miss under miss {g('TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS'):.3e}, stalls: inflight max {g('TCP_UTCL1_STALL_INFLIGHT_MAX'):.3e} "
          f"UTCL2 credits {g('TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS'):.3e} serialisation {g('TCP_UTCL1_SERIALIZATION_STALL'):.3e} thrashing {g('TCP_UTCL1_THRASHING_STALL'):.3e}")
    print(f"   L1: pending stall {g('TCP_PENDING_STALL_CYCLES'):.3e}  TCR stall {g('TCP_TCR_TCP_STALL_CYCLES'):.3e}  active {g('TCP_GATE_EN1'):.3e}  accesses {g('TCP_TOTAL_ACCESSES'):.3e}")
    print(f"   TA: busy {g('TA_TA_BUSY'):.3e}  address stalled by TC {g('TA_ADDR_STALLED_BY_TC_CYCLES'):.3e}  data stalled by TC {g('TA_DATA_STALLED_BY_TC_CYCLES'):.3e}  wavefronts {g('TA_TOTAL_WAVEFRONTS'):.3e}")
    vm = max(g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR"), 1.0)
    print(f"   SQ: vector-memory instructions {vm:.3e} (loads {g('SQ_INSTS_VMEM_RD'):.3e}), mean time outstanding {g('SQ_INST_LEVEL_VMEM') / vm:.0f} cycles; "
          f"TA address FIFO full {g('SQ_VMEM_TA_ADDR_FIFO_FULL'):.3e}  command FIFO full {g('SQ_VMEM_TA_CMD_FIFO_FULL'):.3e}  write-data FIFO full {g('SQ_VMEM_WR_TA_DATA_FIFO_FULL'):.3e}; "
          f"issue cycles loads {g('SQ_INST_CYCLES_VMEM_RD'):.3e} stores {g('SQ_INST_CYCLES_VMEM_WR'):.3e}")
    print(f"   SQ: vector instructions {g('SQ_INSTS_VALU'):.3e}: transcendental f32 {g('SQ_INSTS_VALU_TRANS_F32') / max(g('SQ_INSTS_VALU'), 1):.3f}  int32 {g('SQ_INSTS_VALU_INT32') / max(g('SQ_INSTS_VALU'), 1):.3f}  "
          f"conversions {g('SQ_INSTS_VALU_CVT') / max(g('SQ_INSTS_VALU'), 1):.3f}; SQ_ACTIVE_INST_VALU {g('SQ_ACTIVE_INST_VALU'):.3e}  SQ_BUSY_CU_CYCLES {g('SQ_BUSY_CU_CYCLES'):.3e}  "
          f"SQ_CYCLES {g('SQ_CYCLES'):.3e}  SQ_WAVE_CYCLES {g('SQ_WAVE_CYCLES'):.3e}")
    print(f"   GRBM: active {g('GRBM_GUI_ACTIVE'):.3e}  UTCL2 busy {g('GRBM_UTCL2_BUSY'):.3e}   dispatch time {g('us:' + next(iter(data))) / 1e3:.2f} ms")
