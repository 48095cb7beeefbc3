#!/usr/bin/env python3
"""The encoder's channel-quad ReLU masks (REPO_EPI_MUL_CMASK) alone: enc1 forward with / without writing its mask, the
three data gradients reading the activation vs its mask (2450 frames).    python tools/cmask_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from repo_amd import ops
g=torch.Generator(device='cuda').manual_seed(0)
r=lambda *s,scale=1.0: torch.randn(*s,device='cuda',generator=g)*scale
def timeit(fn,iters=20,warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1)/iters*1e3
n=2450
obs=(torch.rand(n,3,64,64,device='cuda',generator=g)*255).to(torch.uint8)
w0,b0=r(32,3,4,4,scale=0.2),r(32,scale=0.1)
print("enc1 fwd          %.1f us"%timeit(lambda: ops.conv_down(0,obs,w0,b0,epi=ops.EPI_RELU)))
print("enc1 fwd + cmask  %.1f us"%timeit(lambda: ops.conv_down(0,obs,w0,b0,epi=ops.EPI_RELU,want_cmask=True)))
for lay,(cs,hs,cb,hb,k) in {1:(64,14,32,31,4),2:(128,6,64,14,4),3:(256,2,128,6,4)}.items():
    h=torch.relu(r(n,cb,hb,hb)); d=r(n,cs,hs,hs); w=r(cs,cb,k,k,scale=0.1)
    bits=(h>0).to(torch.uint8).view(n,cb//4,4,hb*hb); cm=(bits[:,:,0]|(bits[:,:,1]<<1)|(bits[:,:,2]<<2)|(bits[:,:,3]<<3)).contiguous().view(-1)
    pk=ops.conv_up_pack(lay,w)
    print("layer %d dgrad  relu operand %.1f us   channel-quad mask %.1f us"%(lay,timeit(lambda: ops.conv_up(lay,d,w,None,epi=ops.EPI_MUL_DRELU,aux=h,pack=pk)),timeit(lambda: ops.conv_up(lay,d,w,None,epi=ops.EPI_MUL_CMASK,aux=cm,pack=pk))))
