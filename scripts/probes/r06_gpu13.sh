mkdir -p gpurun_out/r06
V=$PWD/rodygs_amd/csrc/variants/exact_cull.so
# correctness of the variant: image identical, lists shorter
python - <<'PY'
import os, sys, subprocess, json
code = r'''
import sys, torch, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import hip_stages as HS
from oracle import rasterizer_oracle as O
from rodygs_amd import GaussianRasterizer
sc = O.synthetic_scene(1000000, 1920, 1080, 3, seed=777)
hs = HS.run_stages(sc, 3)
r = hs["ranges"].astype(np.int64)
with torch.no_grad():
    out = GaussianRasterizer(HS.make_settings(sc, 3))(means3D=sc["means3D"].cuda(), means2D=torch.zeros(1000000,3).cuda(), shs=sc["shs"].cuda(), opacities=sc["opacities"].cuda(), scales=sc["scales"].cuda(), rotations=sc["rotations"].cuda(), viewmatrix=sc["viewmatrix"].cuda())
print("D", hs["D"], "listed", int((r[:,1]-r[:,0]).sum()), "img", float(out[0].double().sum()), float(out[1].double().sum()))
'''
for lib in ("", os.environ.get("V_LIB","")):
    env = dict(os.environ)
    if lib: env["RDG_LIB_PATH"] = lib
    print(("variant " if lib else "regular ") + subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip())
PY
for i in 1 2 3; do
python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=b['stage_ms']; print('rect only ', 'step %.4f ms' % b['ms_per_step'], 'scan_dup %.1f us' % (1e3*s['scan_dup']), 'sort %.1f' % (1e3*s['sort']), 'render_fwd %.1f' % (1e3*s['render_fwd']), 'render_bwd %.1f' % (1e3*s['render_bwd']))"
RDG_LIB_PATH=$V python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=b['stage_ms']; print('+exact test', 'step %.4f ms' % b['ms_per_step'], 'scan_dup %.1f us' % (1e3*s['scan_dup']), 'sort %.1f' % (1e3*s['sort']), 'render_fwd %.1f' % (1e3*s['render_fwd']), 'render_bwd %.1f' % (1e3*s['render_bwd']))"
done
