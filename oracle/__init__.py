"""CPU oracle package — TEST INFRASTRUCTURE ONLY (see oracle/oracle.cpp header).
Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
package ray_tracing_in_one_weekend_amd never imports it."""
