import sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd']
from models.modules.dct import FAF
faf = FAF().cuda() if hasattr(FAF(), 'cuda') else FAF()
x = torch.randn(8, 5, 3, 224, 224, device='cuda')
for _ in range(3): y = faf.forward_frame(x, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): y = faf.forward_frame(x, 1)
e1.record(); torch.cuda.synchronize()
print(f"faf forward_frame B=8: {e0.elapsed_time(e1) * 1e3 / 20:.1f} us per call, out {tuple(y.shape)}")
