"""cProfile of the host side of the fused step (which Python/ctypes calls the enqueue time goes to)."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dd3d_amd.fused_step as fs
from dd3d_amd.models.MEConv import MEConvExp
from dd3d_amd.synthetic import init_synthetic, mnist3d_like

dev = torch.device("cuda:0")
images, _, idx = mnist3d_like(10, 512, seed=0)
d = fs.FusedDistiller(images, idx, MEConvExp(in_channel=3, out_channel=10), ipc=10, device=dev, seed=1, image_syn=init_synthetic(10, 10, 250, 0))
for _ in range(5):
    d.outer_step()
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(n):
    d.outer_step()
pr.disable()
torch.cuda.synchronize()
print("ms/step (profiled)", (time.perf_counter() - t0) * 1e3 / n)
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(40)
