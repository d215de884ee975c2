"""Time the device-resident minibatch assembly of gemm_gan_amd/data.py at the headline shape (B = 256, 256 of up to 2 000
patches x 1024 per case, 300 x 768 tokens): python tools/feeder_probe.py [n_cases]"""
import sys, time, tempfile, pathlib
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from gemm_gan_amd.data import DeviceCaseCache

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rng = np.random.default_rng(0)
with tempfile.TemporaryDirectory() as d:
    d = pathlib.Path(d); (d / "p").mkdir(); (d / "t").mkdir()
    ids = []
    for i in range(n_cases):
        n = int(rng.integers(50, 2000))
        np.save(d / "p" / f"c{i}.npy", rng.standard_normal((n, 1024)))
        np.save(d / "t" / f"c{i}.npy", rng.standard_normal((1, 300, 768)).astype(np.float32))
        np.save(d / "t" / f"c{i}_attention_mask.npy", np.ones((1, 300), dtype=np.int64))
        ids.append(f"c{i}")
    t0 = time.perf_counter()
    cache = DeviceCaseCache(ids, d / "t", d / "p", rng.standard_normal((n_cases, 5000)), num_patches=256, device="cuda:0")
    torch.cuda.synchronize()
    print(f"cache of {n_cases} cases: {cache.nbytes() / 2**30:.2f} GiB on the device, built in {time.perf_counter() - t0:.1f} s")
loader = cache.loader(batch_size=256, seed=0)
for _ in range(2):
    for b in loader: pass
torch.cuda.synchronize()
t0 = time.perf_counter(); nb = 0
for _ in range(5):
    for b in loader: nb += 1
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / nb * 1e3:.2f} ms per minibatch of 256 (device-side subsample / pad / gather)")
