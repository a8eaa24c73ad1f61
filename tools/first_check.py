import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import oracle_ffi as o
from ec504_imageencoder_amd import Mpeg1Encoder
print(torch.cuda.get_device_name(0))
for (W,H,mode,n) in [(352,288,'full',4),(352,288,'strict',2),(1920,1080,'full',3),(360,250,'full',2)]:
    enc = Mpeg1Encoder(W,H,12,mode,max_frames=8)
    rgb = enc.synth(n, seed=504)
    torch.cuda.synchronize()
    host = o.synth_frames(n,W,H,seed=504)
    print(W,H,mode,'synth equal', np.array_equal(rgb.cpu().numpy(), host))
    m = o.MODE_FULL if mode=='full' else o.MODE_STRICT
    co = enc.coefficients(rgb).cpu().numpy()
    ref = np.stack([o.frame_coefficients(host[f],W,H,12,m) for f in range(n)])
    print('  coeffs equal', np.array_equal(co.astype(np.int32), ref), co.shape)
    got, sizes = enc.encode_to_bytes(rgb, first_frame_index=5)
    want, wsizes = o.encode_frames(host, n, W, H, 5, 12, m, threads=4)
    print('  stream equal', got == want, len(got), len(want), sizes[:3], list(wsizes[:3]))
    if got != want:
        for i,(a,b) in enumerate(zip(got,want)):
            if a!=b: print('   first diff at', i); break
    enc.close()
