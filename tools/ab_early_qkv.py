import sys, time, torch
sys.path.insert(0, '/root/repo')
import wave_mamba_amd as wm, bench
from wave_mamba_amd.archs import wavemamba_arch as arch
dev = torch.device('cuda', 0)
net = bench.build_model(dev)
x = bench.pad_to(torch.rand(1, 3, 2160, 3840, generator=torch.Generator().manual_seed(1234))).to(dev)
unet = net.restoration_network
def t(n=30):
    with torch.no_grad():
        for _ in range(5): unet(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): unet(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    arch.DownFRG.early_qkv = False; a = unet(x).clone()
    arch.DownFRG.early_qkv = True; b = unet(x).clone()
    print('bit-equal:', torch.equal(a, b))
for rep in range(3):
    for flag in (False, True):
        arch.DownFRG.early_qkv = flag
        print(f'early_qkv {flag}: {t():.3f} ms', flush=True)
