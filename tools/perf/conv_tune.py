import sys, math, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from layout2img_amd import ops, _lib
dev = torch.device('cuda:0')
shapes = [(32,128,128,64,64,3,0,0),(32,64,64,64,64,3,1,0),(32,64,64,128,128,3,0,1),(32,32,32,512,512,3,0,0),(256,8,8,512,512,3,0,0),(256,8,8,1024,1024,3,0,1),(256,4,4,1024,1024,3,1,0),(32,16,16,512,512,3,0,0),
          (32,32,32,256,256,3,0,0),(32,64,64,128,128,3,0,0),(32,16,16,512,256,3,0,0),(32,32,32,256,128,3,0,0),(32,64,64,528,104,3,0,0),
          (32,8,8,1024,1024,3,0,1),(32,32,32,128,256,3,0,0),(32,16,16,256,512,3,0,0),(32,8,8,512,512,3,1,0)]
g = torch.Generator().manual_seed(0)
for (B,H,W,Ci,Co,KH,up2,pool2) in shapes:
    x = torch.randn(B,H,W,Ci, generator=g).to(dev, torch.bfloat16)
    K = KH*KH*Ci; kpad = (K+63)//64*64; npad = (Co+127)//128*128
    w = (torch.randn(npad,kpad, generator=g)/math.sqrt(K)).to(dev, torch.bfloat16)
    Ho = H*(2 if up2 else 1)
    res = []
    ref = None
    for cfg in (15,16):
        _lib.call("l2i_set_conv_config", cfg)
        for _ in range(3): out,_,_ = ops.conv_raw(x,w,kpad,Co,KH,up2=bool(up2),pool2=bool(pool2),alpha=0.25 if pool2 else 1.0)
        torch.cuda.synchronize()
        s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): out,_,_ = ops.conv_raw(x,w,kpad,Co,KH,up2=bool(up2),pool2=bool(pool2),alpha=0.25 if pool2 else 1.0)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e)/10
        if ref is None: ref = out.clone()
        err = float((out-ref).abs().max())
        res.append(f"cfg{cfg}:{2.0*B*Ho*Ho*Co*K/ms/1e9:6.0f}TF(err {err:.1e})")
    print((B,H,W,Ci,Co,KH,up2,pool2), "  ".join(res))
_lib.call("l2i_set_conv_config", -1)
