#!/usr/bin/env python3
"""Scratch (CPU, numpy): how often does a wave of the pendulum's implicit filter (C3) pay for the soft saturation's bevel?
A wave executes the bevel's ~25 instructions on every step where ANY of its 64 lanes is between bevelStart and bevelStop.
Simulates the 16 384 seeded backup trajectories and counts (wave, step) pairs with a lane in the bevel for the seeded
order and for lanes dealt by locality (Morton order of the initial state, the initial controller output, and -- as a
bound no key available before the launch reaches -- the step at which the lane first enters the bevel)."""
import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, re
from asif_amd import workloads
src=open('/root/repo/oracle/or_models.c').read()
m=re.search(r'ip_K\[2\]\s*=\s*\{([^}]*)\}',src); K=np.array([float(eval(t)) for t in m.group(1).split(',')])
print('K',K)
B=16384
x,_=workloads.make_batch(3,B,0)   # [2][B]
lb,ub=-1.5,1.5; r=0.1
L=r*np.tan(np.pi/8); start=1-np.cos(np.pi/4)*L; stop=1+L
rng=ub-lb; mid=0.5*(ub+lb)
def simulate(x0):
    x=x0.copy(); N=5001; dt=1e-3
    inb=np.zeros((N,B),dtype=bool)
    for k in range(N):
        u=K[0]*x[0]+K[1]*x[1]
        uc=(u-mid)*2/rng; au=np.abs(uc)
        bev=(au>start)&(au<stop)
        inb[k]=bev
        us=np.clip(u,lb,ub)
        t=au-stop; sq=np.sqrt(np.maximum(r*r-t*t,0)); usb=0.5*(sq+(1-r))*rng
        us=np.where(bev, np.where(uc<0, mid-usb, usb+mid), us)
        x=np.stack([x[0]+dt*x[1], x[1]+dt*(np.sin(x[0])+us)])
    return inb
inb=simulate(x)
print('per-lane bevel fraction of steps', inb.mean())
def wave_rate(order):
    w=inb[:,order].reshape(5001,B//64,64).any(axis=2)
    return w.mean()
print('unsorted: fraction of (wave,step) with a lane in the bevel', wave_rate(np.arange(B)))
def morton(a,b,bits=8):
    qa=np.clip(((a+1.5)/3*(1<<bits)).astype(np.int64),0,(1<<bits)-1); qb=np.clip(((b+1.5)/3*(1<<bits)).astype(np.int64),0,(1<<bits)-1)
    k=np.zeros_like(qa)
    for i in range(bits): k|=((qa>>i)&1)<<(2*i+1); k|=((qb>>i)&1)<<(2*i)
    return k
print('morton(x0,x1):', wave_rate(np.argsort(morton(x[0],x[1]),kind='stable')))
u0=K[0]*x[0]+K[1]*x[1]
print('sorted by u0:', wave_rate(np.argsort(u0,kind='stable')))
# first entry step into bevel as an oracle-ish best key
first=np.where(inb.any(0), inb.argmax(0), 6000)
print('sorted by first bevel step (upper bound-ish):', wave_rate(np.argsort(first,kind='stable')))
print('4x finer: tile 16x16 cells, row-major:', wave_rate(np.lexsort((x[1], np.floor((x[0]+1.5)/3*16)))))
