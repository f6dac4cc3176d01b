import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocha_sigasia2023_amd import ContextBank, Generator, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
model = Generator(device=dev).load_state_dict(synthetic_state_dict(1777, 1.0)).eval()
m_, s_ = synthetic.cnt_norm(7); mean, std = torch.from_numpy(m_).to(dev), torch.from_numpy(s_).to(dev)
cha = torch.from_numpy(synthetic.pose_windows(2, 585)).to(dev)
e, c, n = model.encode(cha, mean, std)
bank = ContextBank(model, n, e)
src_all = torch.from_numpy(synthetic.pose_windows(1, 64)).to(dev)
out = []
for B in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
    src = src_all[:B].contiguous()
    for _ in range(3): bank.characterize(src, mean, std)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): bank.characterize(src, mean, std)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    out.append(f"{B}:{dt*1e3:.3f}")
print(os.environ.get("MOCHA_SKINNY16_MAXM"), " ".join(out))
