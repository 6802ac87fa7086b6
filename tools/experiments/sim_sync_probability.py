#!/usr/bin/env python3
"""CPU simulation (no GPU): how often does a decoder that starts at a chunk boundary with the guess "a block of the MCU's first component
starts here" end its 256-byte chunk in the TRUE state?  Pure-Python Huffman walk over one 1080p 4:2:0 quality-80 bench frame (first 600
chunks); prints the shares of: full agreement / right bit position only / right position and zigzag index but wrong block-in-MCU /
wrong position.  Round 4: {'full': 0.858, 'bp right only': 0.045, 'bp,k right, blk wrong': 0.053, 'bp wrong': 0.043} - which is why the
passes from the third on have only a few per cent of the chunks (k_jpeg_sync_tail).   python tools/experiments/sim_sync_probability.py"""
import io, sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
load_package()
from of_amd import synth
from PIL import Image
p = synth.render_pair(1080,1920,900)
buf=io.BytesIO(); Image.fromarray(p["prev"]).save(buf,"JPEG",quality=80,subsampling=2); d=buf.getvalue()
i=2; huff={}
while True:
    m=d[i+1]; L=(d[i+2]<<8)|d[i+3]; seg=d[i+4:i+2+L]
    if m==0xC4:
        k=0
        while k<len(seg):
            tc,th=seg[k]>>4,seg[k]&15; bits=seg[k+1:k+17]; n=sum(bits); vals=seg[k+17:k+17+n]; k+=17+n
            code=0;p_=0;tab={}
            for l in range(1,17):
                for _ in range(bits[l-1]):
                    tab[(l,code)]=vals[p_]; p_+=1; code+=1
                code<<=1
            huff[(tc,th)]=tab
    if m==0xDA:
        ent=d[i+2+L:-2]; break
    i+=2+L
ent=ent.replace(b'\xff\x00',b'\xff')+bytes(64)
bits=np.unpackbits(np.frombuffer(ent,dtype=np.uint8)).tolist()
def step(pos,k,blk):
    tc=0 if k==0 else 1; th=0 if blk<4 else 1
    tab=huff[(tc,th)]; code=0; sym=None
    for l in range(1,17):
        code=(code<<1)|bits[pos+l-1]
        if (l,code) in tab: sym=tab[(l,code)]; ln=l; break
    if sym is None: sym=0; ln=16
    s=sym&15; r=(sym>>4) if k else 0
    pos+=ln+s
    if s: k+=r+1
    elif k==0: k=1
    elif r==15: k+=16
    else: k=64
    if k>=64: k=0; blk=(blk+1)%6
    return pos,k,blk
CH=256*8
nbits=(len(ent)-64)*8
nch=nbits//CH
# truth: state at first symbol start >= chunk boundary
truth={}
pos,k,blk=0,0,0; c=1
while pos<nbits and c<=nch:
    if pos>=c*CH:
        truth[c]=(pos,k,blk); c+=1; continue
    pos,k,blk=step(pos,k,blk)
import collections
res=collections.Counter()
N=min(nch-1,600)
for c in range(1,N):
    pos,k,blk=c*CH,0,0
    while pos<(c+1)*CH: pos,k,blk=step(pos,k,blk)
    t=truth[c+1]
    if (pos,k,blk)==t: res["full"]+=1
    elif pos==t[0] and k==t[1]: res["bp,k right, blk wrong"]+=1
    elif pos==t[0]: res["bp right only"]+=1
    else: res["bp wrong"]+=1
print({a:round(b/(N-1),3) for a,b in res.items()})
