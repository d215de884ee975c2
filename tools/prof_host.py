"""Which host-side torch ops surround one WGAN_GP.train() (tools only): torch.profiler table of op counts."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gemm_gan_amd as gga
dev = torch.device("cuda:0")
G, B, P, T = 5000, 256, 256, 1
torch.manual_seed(42)
w = gga.WGAN_GP(G, 256, 256, [256, 256, G], [256, 256, 1], text_embedding_dims=512, patches_embedding_dims=1024,
                optimizer="rms_prop", n_critic=5, dropout=0.1, seed=1, device=dev, results_dire="", precision="bf16")
w.build_WGAN_GP(); w.init_train(); w.reserve(B, P, T)
x = torch.randn(B, G, device=dev); patches = torch.randn(B, P, 1024, device=dev); text = torch.randn(B, T, 512, device=dev)
pp = torch.zeros(B, P, dtype=torch.bool, device=dev); tp = torch.zeros(B, T, dtype=torch.bool, device=dev)
for _ in range(2): w.train(x, text, tp, patches, pp)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(2): w.train(x, text, tp, patches, pp)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=25, max_name_column_width=60))
