import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracing_in_one_weekend_amd as rt
rt.register_default_images()
scene = rt.Scene.build("sphere_scene", 16/9)
r = rt.Renderer(0); r.upload(scene)
p = rt.make_params(1920, 1080, 256, max_depth=50)
a,_,sa = r.render(scene.camera, p)
r.set_option("primary_lists", 1)
b,_,sb = r.render(scene.camera, p)
r.set_option("primary_lists", 0)
pb = rt.make_params(1920, 1080, 256, max_depth=50, flags=rt._ffi.FLAG_BRUTE_FORCE)
c,_,sc = r.render(scene.camera, pb)
print("lists", sa.n_rays, "tree", sb.n_rays, "brute", sc.n_rays)
for name,x in (("lists",a),("tree",b)):
    d = (x.view(np.uint32) != c.view(np.uint32)).any(axis=2)
    print(name, "pixels differing from brute force:", int(d.sum()), np.argwhere(d)[:5].tolist())
print("depth0..3 lists", list(sa.rays_per_depth[:4]), "tree", list(sb.rays_per_depth[:4]), "brute", list(sc.rays_per_depth[:4]))
