"""Host-side C++ mirror of the reference constructors (librtow_host.so) against the oracle's
independent restatement of demo_scene.rs / texture.rs / camera.rs."""
import ctypes as C

import numpy as np
import pytest

F = C.POINTER(C.c_float)


def test_sphere_scene_layout_matches_oracle_restatement(rt, orc):
    lib = orc.load()
    s = rt.Scene.build("sphere_scene", 16 / 9)
    a = s.arrays()
    assert len(a["sph_r"]) == 533  # 4 + 23*23, no rejection around the big spheres (demo_scene.rs:58-77)
    assert [s.sphere_name(i) for i in range(4)] == ["Ground", "Sphere_1", "Sphere_2", "Sphere_3"]
    assert s.sphere_name(4) == "Sphere -11, -11"
    assert a["sph_r"][:4].tolist() == [1000.0, 1.0, 1.0, 1.0] and np.all(a["sph_r"][4:] == np.float32(0.2))
    lay = np.zeros(529 * 8, np.float32)
    assert lib.orc_sphere_scene_layout(95, lay.ctypes.data_as(F)) == 529
    lay = lay.reshape(529, 8)
    assert np.array_equal(a["sph_cx"][4:], lay[:, 0]) and np.array_equal(a["sph_cz"][4:], lay[:, 2])
    assert np.all(a["sph_cy"][4:] == np.float32(0.2))
    mt = a["mat_type"][a["sph_mat"][4:]]
    kinds = lay[:, 3].astype(int)
    assert np.array_equal(mt == rt._ffi.MAT_DIFFUSE, kinds == 0)
    assert np.array_equal(mt == rt._ffi.MAT_METAL, kinds == 1)
    assert np.array_equal(mt == rt._ffi.MAT_DIELECTRIC, kinds == 2)
    # colours: Diffuse -> ConstantTex colour, Metal -> albedo + fuzz
    for i in range(529):
        m = a["sph_mat"][4 + i]
        if kinds[i] == 0:
            t = a["mat_tex0"][m]
            assert a["tex_type"][t] == rt._ffi.TEX_CONSTANT
            assert np.array_equal(a["tex_color0"][3 * t:3 * t + 3], lay[i, 4:7])
        elif kinds[i] == 1:
            assert np.array_equal(a["mat_color"][3 * m:3 * m + 3], lay[i, 4:7]) and a["mat_p0"][m] == lay[i, 7]
    # the shared Dielectric Arc stays one material (demo_scene.rs:46,52,73)
    glass = {int(a["sph_mat"][4 + i]) for i in range(529) if kinds[i] == 2}
    assert glass == {int(a["sph_mat"][2])}
    assert a["sky_type"] == rt._ffi.SKY_GRADIENT


def test_perlin_tables_follow_thread_rng_seed_1995(rt, orc):
    lib = orc.load()
    rt._ffi.load_host_library().rth_rng_reseed(1995)
    s = rt.Scene.build("sphere_scene", 16 / 9)
    a = s.arrays()
    vec = np.zeros(768, np.float32)
    perm = np.zeros(768, np.uint16)
    lib.orc_perlin_tables(1995, 1, vec.ctypes.data_as(F), perm.ctypes.data_as(C.POINTER(C.c_uint16)))
    assert np.array_equal(a["perlin_vec"], vec) and np.array_equal(a["perlin_perm"], perm)
    for k in range(3):
        assert sorted(perm[256 * k:256 * (k + 1)].tolist()) == list(range(256))
    assert np.all(np.abs(vec) <= 1.0)


def test_camera_matches_oracle(rt, orc):
    lib = orc.load()
    s = rt.Scene.build("sphere_scene", 400 / 225)
    cam = rt.RtCamera()
    f = lambda v: np.asarray(v, np.float32).ctypes.data_as(F)
    lib.orc_camera_new(f([13, 2, 3]), f([0, 0, 0]), f([0, 1, 0]), 20.0, C.c_float(400 / 225), C.byref(cam))
    for name in ("origin", "horizontal", "vertical", "lower_left_corner"):
        assert list(getattr(cam, name)) == list(getattr(s.camera, name)), name


def test_test_sphere_scene(rt):
    s = rt.Scene.build("test_sphere", 2.0)
    a = s.arrays()
    assert len(a["sph_r"]) == 2 and a["mat_type"].tolist() == [rt._ffi.MAT_LAMBERT]  # shared `ground` Arc
    assert a["sph_cy"].tolist() == [-100.5, 0.0] and a["sph_r"].tolist() == [100.0, 0.5]
    assert list(s.camera.lower_left_corner) == [-2.0, -1.0, -1.0]


def test_build_authored_scenes(rt):
    e = rt.Scene.build("earth_env_scene", 16 / 9)
    assert e.flat.sky_type == rt._ffi.SKY_ENV and e.flat.n_images == 2 and e.flat.n_spheres == 5
    p = rt.Scene.build("pbr_sweep_scene", 16 / 9)
    a = p.arrays()
    assert p.flat.n_spheres == 501
    assert set(a["mat_type"].tolist()) >= {rt._ffi.MAT_DISNEY_METAL, rt._ffi.MAT_ROUGH_PLASTIC, rt._ffi.MAT_DISNEY_CLEARCOAT,
                                            rt._ffi.MAT_OREN_NAYAR, rt._ffi.MAT_BURLEY_DIFFUSE, rt._ffi.MAT_DISNEY_DIFFUSE,
                                            rt._ffi.MAT_DISNEY_SHEEN}


def test_piecewise_construction_and_errors(rt):
    s = rt.Scene.new()
    t = s.constant_tex((0.1, 0.2, 0.3))
    m = s.material(rt._ffi.MAT_DIFFUSE, tex0=t)
    s.sphere((0, 0, -1), 0.5, m, "a")
    s.sphere((0, 1, -1), 0.25, m, "b")
    s.set_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 90, 2.0)
    s.finish()
    assert s.flat.n_spheres == 2 and s.flat.n_materials == 1 and s.flat.n_textures == 1
    with pytest.raises(rt.RtError):
        rt.Scene.build("no_such_scene", 1.0)
    s2 = rt.Scene.new()
    with pytest.raises(rt.RtError):
        s2.material(rt._ffi.MAT_DIFFUSE)  # Diffuse needs an albedo texture
    with pytest.raises(rt.RtError):
        s2.image_tex("res/does_not_exist.jpg")  # image::open(..).unwrap() panics in the reference
    with pytest.raises(rt.RtError):
        s2.finish()  # no camera


def test_simple_light_scene_and_boxes(rt):
    s = rt.Scene.build("simple_light_scene", 16 / 9)  # demo_scene.rs:88-110
    a = s.arrays()
    assert s.flat.n_spheres == 3 and s.flat.n_rects == 1 and s.flat.sky_type == rt._ffi.SKY_BLACK
    assert a["rect_axis"].tolist() == [rt._ffi.RECT_XY] and a["rect_min"].tolist() == [3, 1, -2] and a["rect_max"].tolist() == [5, 3, -2]
    # the emissive material is shared by Sphere_2 and the rect; the Perlin material by Ground and Sphere_1
    assert a["sph_mat"].tolist() == [0, 0, 1] and a["rect_mat"].tolist() == [1]
    assert a["mat_type"].tolist() == [rt._ffi.MAT_DIFFUSE, rt._ffi.MAT_EMISSION]
    b = rt.Scene.new()
    m = b.material(rt._ffi.MAT_DIFFUSE, tex0=b.constant_tex((0.5, 0.5, 0.5)))
    b.gbox((0, 0, 0), (1, 2, 3), m)
    b.set_camera((5, 5, 5), (0, 0, 0), (0, 1, 0), 40, 1.0)
    b.finish()
    a = b.arrays()  # GBox::new order: XY(min.z), XY(max.z), XZ(min.y), XZ(max.y), YZ(min.x), YZ(max.x)
    assert a["rect_axis"].tolist() == [2, 2, 1, 1, 0, 0]
    assert a["rect_min"].reshape(6, 3)[[0, 1, 2, 3, 4, 5], [2, 2, 1, 1, 0, 0]].tolist() == [0, 3, 0, 2, 0, 1]


def test_cornell_box_mirror(rt):
    s = rt.Scene.build("cornell_box", 1.0)  # demo_scene.rs:112-148
    a = s.arrays()
    assert s.flat.n_spheres == 0 and s.flat.n_rects == 18 and s.flat.n_media == 2 and s.flat.n_xforms == 4
    assert a["xf_type"].tolist() == [0, 1, 0, 1] and a["xf_param"].reshape(4, 4)[0, :3].tolist() == [265.0, 0.0, 295.0]
    assert a["xf_param"].reshape(4, 4)[1, 2] == 15.0 and a["xf_param"].reshape(4, 4)[3, 2] == -18.0
    assert a["rect_xform"].tolist() == [rt._ffi.NO_XFORM] * 6 + [1] * 6 + [3] * 6
    assert a["med_neg_inv_density"].tolist() == [-100.0, -100.0]  # -1 / 0.01
    assert [a["mat_type"][m] for m in a["med_mat"]] == [rt._ffi.MAT_ISOTROPIC] * 2
    # phase textures: black smoke, white smoke (demo_scene.rs:123,128)
    t0, t1 = a["mat_tex0"][a["med_mat"][0]], a["mat_tex0"][a["med_mat"][1]]
    assert a["tex_color0"][3 * t0:3 * t0 + 3].tolist() == [0, 0, 0] and a["tex_color0"][3 * t1:3 * t1 + 3].tolist() == [1, 1, 1]


def test_final_scene_mirror(rt, orc):
    """demo_scene.rs:150-221: 1000-sphere instanced cloud + 5 spheres + the world copy of `boundary` + its medium copy,
    400 boxes + the light, RotateY(15) under Translate, one ConstantMedium(0.2) with a sphere boundary."""
    s = rt.Scene.build("final_scene", 1.0)
    a = s.arrays()
    fs = s.flat
    assert (fs.n_spheres, fs.n_rects, fs.n_xforms, fs.n_media, fs.n_materials) == (1007, 2401, 2, 1, 9)
    assert fs.sky_type == rt._ffi.SKY_BLACK
    # flatten order follows the world vector: light rect first, cloud spheres first among the spheres
    rmin = a["rect_min"].reshape(-1, 3)
    assert a["rect_axis"][0] == rt._ffi.RECT_XZ and rmin[0].tolist() == [123.0, 544.0, 147.0]
    cloud = np.stack([a["sph_cx"][:1000], a["sph_cy"][:1000], a["sph_cz"][:1000]], axis=1)
    assert (cloud >= 0).all() and (cloud < 165).all() and (a["sph_r"][:1000] == 10).all()
    assert (a["sph_xform"][:1000] == a["sph_xform"][0]).all() and a["sph_xform"][0] != rt._ffi.NO_XFORM
    assert (a["sph_xform"][1000:] == rt._ffi.NO_XFORM).all() and (a["rect_xform"] == rt._ffi.NO_XFORM).all()
    chain = a["sph_xform"][0]      # innermost wrapper first: RotateY, whose parent is the Translate
    assert a["xf_type"][chain] == rt._ffi.XF_ROTATE_Y and a["xf_type"][a["xf_parent"][chain]] == rt._ffi.XF_TRANSLATE
    assert a["xf_param"].reshape(-1, 4)[a["xf_parent"][chain]][:3].tolist() == [-100.0, 270.0, 395.0]
    # `boundary` appears twice: as a glass sphere of the world and as the medium's boundary
    assert a["sph_medium"].tolist() == [rt._ffi.NO_XFORM] * 1006 + [0]
    assert [a[k][1005] for k in ("sph_cx", "sph_cy", "sph_cz", "sph_r")] == [a[k][1006] for k in ("sph_cx", "sph_cy", "sph_cz", "sph_r")]
    assert a["med_neg_inv_density"].tolist() == [-5.0]
    # thread-RNG order: Perlin tables first (768 floats + 3 shuffles), then the cloud, then the box heights
    want_c = np.zeros(3000, np.float32)
    want_h = np.zeros(400, np.float32)
    orc.load().orc_final_scene_layout(1995, orc._fp(want_c), orc._fp(want_h))
    assert np.array_equal(cloud.view(np.uint32), want_c.reshape(1000, 3).view(np.uint32))
    heights = rmin[1:][3::6, 1]   # side 3 of every GBox is the top XZRect, whose plane is y = max.y (hitable.rs:372-379)
    assert heights.shape == (400,) and np.array_equal(heights.view(np.uint32), want_h.view(np.uint32))
    assert (heights >= 1).all() and (heights < 101).all()


def test_bvh_shape_decides_medium_visit_count(rt):
    """hitable.rs:177-221: a span of 3 splits 1 + 2 after the sort by box min, so the object with the smallest
    coordinates sits alone in a node (left == right) and BvhNode::hit calls it twice; for a ConstantMedium that
    doubles the scatter rate.  Positions are ordered the same on every axis, so the axis draw does not matter;
    construction order is shuffled to show that the sort, not the order, decides."""
    s = rt.Scene.new()
    glass = s.material(rt._ffi.MAT_DIELECTRIC, p=(1.5, 0, 0, 0))
    tex = s.constant_tex((1, 1, 1))
    for c in ((0, 0, 0), (3, 3, 3), (-3, -3, -3)):
        s.constant_medium(s.sphere(c, 1.0, glass, "b"), 0.5, tex)
    s.set_camera((0, 0, 20), (0, 0, 0), (0, 1, 0), 40, 1.0)
    s.finish(use_bvh=True)
    a = s.arrays()
    assert a["sph_cx"].tolist() == [0.0, 3.0, -3.0]                     # flatten keeps construction order
    assert a["med_neg_inv_density"].tolist() == [-2.0, -2.0, -1.0]      # only the ball at (-3,-3,-3) is alone
    s = rt.Scene.new()
    glass = s.material(rt._ffi.MAT_DIELECTRIC, p=(1.5, 0, 0, 0))
    tex = s.constant_tex((1, 1, 1))
    for c in ((0, 0, 0), (3, 3, 3), (-3, -3, -3)):
        s.constant_medium(s.sphere(c, 1.0, glass, "b"), 0.5, tex)
    s.set_camera((0, 0, 20), (0, 0, 0), (0, 1, 0), 40, 1.0)
    s.finish(use_bvh=False)                                              # a plain HitableList visits each object once
    assert s.arrays()["med_neg_inv_density"].tolist() == [-2.0, -2.0, -2.0]


def test_png_writer_and_output_name(rt, tmp_path):
    """main.rs:110-112,121,127-128: 8-bit RGB PNG with the pixels as given, and the time-stamped file name."""
    from PIL import Image
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    path = tmp_path / "out.png"
    rt.save_png(path, img)
    back = Image.open(path)
    assert back.mode == "RGB" and back.size == (53, 37)
    assert np.array_equal(np.asarray(back), img)
    assert not (tmp_path / "out.png.part").exists()
    with pytest.raises(ValueError):
        rt.save_png(path, img[:, :, 0])
    import re
    import time
    name = rt.output_file_name()
    assert re.fullmatch(r"\d{4}-\d{2}-\d{2}T\d{2}-\d{2}-\d{2}\.png", name)
    t = 1715972624  # any fixed instant: the name is the local wall-clock time of it
    lt = time.localtime(t)
    assert rt.output_file_name(t) == time.strftime("%Y-%m-%dT%H-%M-%S.png", lt)


def test_bbox_of_wrapped_hitables_matches_the_oracle(rt, orc):
    """Two independent restatements of Hitable::bbox (the host mirror's virtual methods, the oracle's flat-scene
    functions) through random Translate / RotateY chains: hitable.rs:104-108, 274-278, 420-431, 449-474, 581-583.
    BvhNode::new sorts by these boxes, so they decide the tree shape the mirror reproduces."""
    rng = np.random.default_rng(11)
    s = rt.Scene.new()
    m = s.material(rt._ffi.MAT_DIFFUSE, tex0=s.constant_tex((0.5, 0.5, 0.5)))
    handles = []
    for i in range(60):
        if i % 2 == 0:
            h = s.sphere(tuple(rng.uniform(-50, 50, 3)), float(rng.uniform(0.1, 9)), m, "s")
        else:
            mn = rng.uniform(-50, 40, 3)
            h = s.rect(int(rng.integers(3)), tuple(mn), tuple(mn + rng.uniform(0.1, 20, 3)), m)
        for _ in range(int(rng.integers(0, 4))):
            h = s.translate(h, tuple(rng.uniform(-30, 30, 3))) if rng.random() < 0.5 else s.rotate_y(h, float(rng.uniform(-180, 180)))
        if i % 7 == 0:
            s.constant_medium(h, 0.5, s.constant_tex((1, 1, 1)))
        handles.append(h)
    boxes = [s.bbox(h) for h in handles]
    s.set_camera((0, 0, 100), (0, 0, 0), (0, 1, 0), 40, 1.0)
    s.finish(use_bvh=False)
    fs = s.flat
    lib = orc.load()
    n_prims = fs.n_spheres + fs.n_rects
    a = s.arrays()
    # construction order: spheres are entries 0.., rects n_spheres..; a medium's box is its boundary's (hitable.rs:581-583)
    si = ri = 0
    for i, (has, box) in enumerate(boxes):
        idx = si if i % 2 == 0 else fs.n_spheres + ri
        si, ri = si + (i % 2 == 0), ri + (i % 2 == 1)
        want = np.zeros(6, np.float32)
        lib.orc_entry_bbox(s.flat_ptr, idx, orc._fp(want))
        assert has and np.array_equal(box.view(np.uint32), want.view(np.uint32)), (i, box, want)
        med = (a["sph_medium"][idx] if idx < fs.n_spheres else a["rect_medium"][idx - fs.n_spheres])
        if med != rt._ffi.NO_XFORM:
            lib.orc_entry_bbox(s.flat_ptr, n_prims + int(med), orc._fp(want))
            assert np.array_equal(box.view(np.uint32), want.view(np.uint32))
