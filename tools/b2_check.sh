mkdir -p gpurun_out/r4j
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_eval_caller.py tests/test_gpu_sharded.py -m gpu -q -s -k "batch_of_two or eval or sharded or exchange or step_small" 2>&1 | grep -E "parity|property|passed|failed|Error|error|assert|FAILED|^E " | tail -30
python - <<'PY'
import torch, time, sys
sys.path.insert(0, ".")
from tests import golden_inputs as gi
from tests.test_gpu_model import make_model, to_dev
from morphablediffusion_amd import synthetic
from morphablediffusion_amd.spec import VolumeConfig
N = 16
m = make_model(gi.FULL_UNET, VolumeConfig(num_views=N), N, workspace_gb=60.0)
bs = [synthetic.make_batch(N, "perspective", 5023, mesh_seed=1 + i) for i in range(2)]
nv = min(b["vertices"].shape[1] for b in bs)
from morphablediffusion_amd.batch import voxelize
def cut(b):
    v = b["vertices"][:, :nv]; coord, out_sh, bounds = voxelize(v[0])
    return dict(b, vertices=v, coord=coord[None], out_sh=out_sh[None], bounds=bounds[None])
bs = [cut(b) for b in bs]
both = to_dev({k: torch.cat([b[k] for b in bs]) for k in bs[0]})
g = torch.Generator().manual_seed(4)
x = torch.randn(2, N, 4, 32, 32, generator=g).cuda(); x_in = (torch.randn(2, 4, 32, 32, generator=g) * 0.18215).cuda()
clip = torch.randn(2, 1, 768, generator=g).cuda(); noise = torch.randn(2, N, 4, 32, 32, generator=g).cuda()
index = 30
ts = torch.full((2,), int(m.sampler.ddim_timesteps[index]), dtype=torch.long, device="cuda")
hs = [int(m.sampler.ddim_timesteps[index])] * 2
def run(B, mode):
    m.sampler.sample_batching = mode
    f = lambda: m.sampler.denoise_apply(x[:B], {"x": x_in[:B]}, clip[:B], ts[:B], index, 2.0, batch_view_num=N, batch={k: v[:B] for k, v in both.items()}, noise=noise[:B], host_steps=hs[:B])
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 10 * 1e3
t1 = run(1, "loop"); tl = run(2, "loop"); tb = run(2, "batched")
print(f"[property] full width N=16: B=1 step {t1:.2f} ms; B=2 looped {tl:.2f} ms ({tl/t1:.2f}x); B=2 batched {tb:.2f} ms ({tb/t1:.2f}x)")
PY
