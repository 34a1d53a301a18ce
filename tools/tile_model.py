#!/usr/bin/env python3
"""CPU model of the list-driven RoI-pool backward's tiling on the fixed roofline RoI set (no GPU needed):
for a tile shape TH x TW, how many (bin, tile) slots the walk visits (`f` = slots / bins: the factor by which
bins that straddle tile borders are read more than once), the slot count of the heaviest tile (the longest
chain one wave walks alone) and how well a longest-first schedule of (tile, channel group) waves fills the
chip's wave slots (`eff` = total work / (slots x makespan)).  DESIGN.md section 4 "Round 4: the floor of the
exact walk" quotes these numbers.

    python3 tools/tile_model.py          -> profiles/r04_tile_model.txt has the output
"""
import os
import heapq

import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = np.load(os.path.join(ROOT, 'profiles', 'roofline_rois_r8512.npy'))
H,W=38,63
def windows(roi):
    # CUDA rounding, forward windows (roi_pooling_op_gpu.cu.cc:40-64)
    x1,y1,x2,y2=[int(np.floor(v*0.0625+0.5)) for v in roi[1:]]
    rw=max(x2-x1+1,1); rh=max(y2-y1+1,1)
    bw=np.float32(rw)/np.float32(7); bh=np.float32(rh)/np.float32(7)
    hs=[min(max(int(np.floor(np.float32(p)*bh))+y1,0),H) for p in range(7)]
    he=[min(max(int(np.ceil(np.float32(p+1)*bh))+y1,0),H) for p in range(7)]
    ws=[min(max(int(np.floor(np.float32(p)*bw))+x1,0),W) for p in range(7)]
    we=[min(max(int(np.ceil(np.float32(p+1)*bw))+x1,0),W) for p in range(7)]
    return hs,he,ws,we
wins=[windows(x) for x in r]
def tile_slots(TH,TW):
    th=-(-H//TH); tw=-(-W//TW)
    cnt=np.zeros((8,th,tw),np.int64)
    for roi,(hs,he,ws,we) in zip(r,wins):
        n=int(roi[0])
        rowc=np.zeros(th,np.int64); colc=np.zeros(tw,np.int64)
        for p in range(7):
            if he[p]>hs[p]:
                for t in range(hs[p]//TH,(he[p]-1)//TH+1): rowc[t]+=1
            if we[p]>ws[p]:
                for t in range(ws[p]//TW,(we[p]-1)//TW+1): colc[t]+=1
        cnt[n]+=np.outer(rowc,colc)
    return cnt
def lpt(work,slots):
    h=[0]*slots; heapq.heapify(h)
    for w in sorted(work,reverse=True):
        heapq.heappush(h,heapq.heappop(h)+w)
    return max(h)
for TH,TW,wpc,groups in [(6,6,8,8),(8,8,8,16),(12,13,8,32),(10,16,8,32),(13,16,6,32),(16,16,5,32),(13,21,4,32),(19,16,4,32),(16,21,4,32),(19,21,3,32),(19,32,2,32),(8,11,8,32),(10,11,8,32),(7,11,8,32),(10,13,8,32),(12,11,8,32)]:
    c=tile_slots(TH,TW)
    tot=c.sum(); f=tot/(8512*49)
    work=np.repeat(c.reshape(-1),groups)
    slots=256*wpc
    # time units: slots processed per wave; 128ch wave: 1 slot/step, 32ch: 4 slots/step
    ms=lpt(work.tolist(),slots)
    print(TH,TW,'wpc',wpc,'f=%.3f'%f,'waves',len(work),'max tile',c.max(),'total/slots=%.0f'%(work.sum()/slots),'lpt makespan=%d'%ms,'eff=%.2f'%(work.sum()/slots/ms))
