import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import misonet_amd as mz
from misonet_amd import weights as W
sd1 = W.make_state_dict(W.miso1_spec(), 0)
m = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0); m.load_state_dict(sd1); m.eval()
g = np.load('/root/repo/tests/golden/g1_miso1_T32.npz')
x = torch.from_numpy(g['x']).cuda()
y = m(x)
t = m.tap("dec6", 1, 32).cpu().numpy()      # [1,4,T,F]
np.save(f"/root/repo/gpurun_out/dec6_rm{os.environ.get('MISONET_X6_RM','1')}.npy", t)
