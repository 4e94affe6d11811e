"""rocprofv3 target: the fused PPO gradient at one minibatch size, with and without the row index."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_roboy_amd.ppo import FusedPolicyGrad, MlpPolicy
B, od, ad = 8388608, 9, 8
policy = MlpPolicy(od, ad).cuda()
fg = FusedPolicyGrad(policy)
N = 4 * B
mb = [torch.rand(N, od, device="cuda"), torch.randn(N, ad, device="cuda") * 0.5, torch.randn(B, device="cuda"),
      torch.randn(N, device="cuda") - 8.0, torch.randn(N, device="cuda"), torch.randn(N, device="cuda")]
perm = torch.randperm(N, device="cuda")[:B].contiguous()
for _ in range(3):
    fg.run(*mb, 0.2, 0.5, 0.1, index=perm)
torch.cuda.synchronize()
small = [t[:B].contiguous() for t in mb]
for _ in range(3):
    fg.run(*small, 0.2, 0.5, 0.1)
torch.cuda.synchronize()
