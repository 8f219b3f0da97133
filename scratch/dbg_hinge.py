import torch, sys
sys.path.insert(0, '.')
from layout2img_amd import _lib
dev='cuda:0'
g = torch.Generator().manual_seed(6)
x = (torch.randn(37, generator=g) * 2).to(dev)
for mode in (0,1,2):
    loss = torch.zeros(4, device=dev); grad = torch.full((40,), -7.0, device=dev)
    torch.cuda.synchronize()
    _lib.call("l2i_hinge_fwd_bwd", x.data_ptr(), None, 37, mode, 0.7, None, loss.data_ptr(), grad.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    xc = x.cpu()
    ref = [torch.relu(1-xc).mean(), torch.relu(1+xc).mean(), -xc.mean()][mode]*0.7
    print(mode, loss.tolist(), float(ref), grad[:4].tolist(), grad[36:].tolist())
z = torch.zeros((), device=dev); print('zeros0d', float(z), z.data_ptr() % 16)
