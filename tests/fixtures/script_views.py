# Test fixture: a data-generation script of the kind users write against the reference's embedded Python module.  Its only renderer import is
# `import volpy`; it is started unmodified by `volren script_views.py --render -w W -h H` or `python -m volren_amd.run_script ...`
# (tests/test_gpu_parity.py::test_scripts_written_for_volpy_run_unmodified).  Inputs come from the environment so that the file itself stays fixed.
import json
import math
import os
import sys

import numpy as np

import volpy

if __name__ == "__main__":
    here = os.path.dirname(__file__)
    out_dir = os.environ["VIEWS_OUT"]
    n_views, spp = int(os.environ.get("VIEWS_N", "2")), int(os.environ.get("VIEWS_SPP", "4"))

    renderer = volpy.Renderer()                       # its size is the -w / -h of the command line
    renderer.init()
    renderer.draw()
    renderer.seed = 7
    renderer.bounces = 12
    renderer.volume = volpy.Volume(os.path.join(here, "smoke.brick"))
    renderer.albedo = volpy.vec3(0.8, 0.85, 0.9)
    renderer.phase = 0.25
    renderer.density_scale = 0.5
    renderer.environment = volpy.Environment(os.path.join(here, "table_mountain_2_puresky_1k.hdr"))
    renderer.environment.strength = 1.5
    renderer.show_environment = True
    renderer.tonemapping = True
    renderer.scale_and_move_to_unit_cube()
    renderer.commit()

    size = renderer.resolution()
    views = []
    for i in range(n_views):
        bb_min, bb_max = renderer.volume.AABB("density")
        center = bb_min + (bb_max - bb_min) * 0.5
        radius = (bb_max - center).length()
        a = 2.0 * math.pi * (i + 0.25) / n_views
        renderer.cam_pos = center + volpy.vec3(math.cos(a), 0.3, math.sin(a)) * (radius * 1.5)
        renderer.cam_dir = (center - renderer.cam_pos).normalize()
        renderer.cam_fov = 45 + 5 * i
        renderer.render(spp)
        renderer.draw()
        np.save(os.path.join(out_dir, "view_%03d.npy" % i), np.array(renderer.fbo_data()))
        renderer.save_with_alpha(os.path.join(out_dir, "view_%03d.png" % i))
        views.append(dict(cam_pos=[float(c) for c in np.array(renderer.cam_pos)], cam_dir=[float(c) for c in np.array(renderer.cam_dir)],
                          cam_fov=float(renderer.cam_fov), density_scale=float(renderer.density_scale)))
    with open(os.path.join(out_dir, "views.json"), "w") as f:
        json.dump(dict(width=int(size.x), height=int(size.y), spp=spp, argv=sys.argv[1:], views=views), f)
    renderer.shutdown()
