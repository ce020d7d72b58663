"""Simulation of the Radon kernel's LDS bank conflicts for tile layouts (companion of radon_lds_bank_sim.py).
A half-wave pass of ds_read2_b32 serves 32 lanes = 2 adjacent angles x 16 adjacent distance bins: two "combs" of 16 points
spaced 1.886 px along the line normal that nearly coincide.  Result (mean conflict factor over line directions, both
streams of the derivative): 97/95 rule 2.58, XOR swizzles 3.0-3.4, "bank = index along the axis the normal is closer
to" (row stride 96, tile transposed for y-normals) exactly 2.0 -- each comb conflict-free in itself, the two combs 2-way.
Measured on the device: 0.771 -> 0.699 ms per 1024^2 image.  python scripts/analysis/radon_lds_layouts.py"""
import numpy as np
rng=np.random.default_rng(1)
n_alpha=n_t=768; W=H=1024; D=np.sqrt(2)*1024
def cycles(words):
    b=words%32; c=0
    for bank in np.unique(b):
        c=max(c,len(np.unique(words[b==bank])))
    return c
def sim_dir(theta, addr, arr, trials=40, both_in_pass=False):
    tot=0; ideal=0
    for _ in range(trials):
        ia0=int((theta/np.pi+0.5)*n_alpha)//16*16; it0=rng.integers(8,n_t//32-8)*32
        w=rng.integers(0,4)
        a_idx,t_idx=arr(ia0,it0,w)
        alpha=(a_idx/n_alpha-0.5)*np.pi; tau=(t_idx/n_t-0.5)*D
        l0,l1=-np.sin(alpha),np.cos(alpha)
        l2=-tau - 0.5*W*l0 - 0.5*H*l1
        ox,oy=-l2*l0,-l2*l1; dx,dy=l1,-l0
        U=rng.uniform(-300,300); t=U+rng.uniform(0,0.66,64)
        nx,ny=l0.mean(),l1.mean()
        for i in range(6):
            for (sx,sy) in ((0,0),(1,1)):
                x=ox+(t+0.66*i)*dx+0.5+sx*dy; y=oy+(t+0.66*i)*dy+0.5-sx*dx
                fi=np.floor(x-0.5).astype(int)+200; fj=np.floor(y-0.5).astype(int)+200
                for rr in (0,1):
                    w0=addr(fi,fj+rr,nx,ny); w1=addr(fi+1,fj+rr,nx,ny)
                    for h in (0,32):
                        if both_in_pass:
                            for q in (0,16):
                                tot+=cycles(np.concatenate([w0[h+q:h+q+16],w1[h+q:h+q+16]])); ideal+=1
                        else:
                            tot+=cycles(w0[h:h+32])+cycles(w1[h:h+32]); ideal+=2
    return tot/ideal
cur=lambda ia0,it0,w:(ia0+4*w+np.repeat(np.arange(4),16), it0+np.tile(np.arange(16),4))
a32=lambda ia0,it0,w:(ia0+2*w+np.repeat(np.arange(2),32), it0+np.tile(np.arange(32),2))
def lin(S):
    return lambda fi,fj,nx,ny: fj*S+fi
def rule(fi,fj,nx,ny):
    S=97 if abs(nx+ny)>=abs(nx-ny) else 95
    return fj*S+fi
def xor_sw(k,S=96):
    return lambda fi,fj,nx,ny: fj*S + (fi ^ ((fj*k)&31))
def transp(fi,fj,nx,ny):
    # bank = x mod 32 for x-major normals, y mod 32 otherwise
    return fj*96+fi if abs(nx)>=abs(ny) else fi*96+fj
def rot(k):
    return lambda fi,fj,nx,ny: fj*128 + ((fi + k*fj)&127)
thetas=np.linspace(-np.pi/2+0.02,np.pi/2-0.02,12)
def mean(addr,arr,**kw): return np.mean([sim_dir(t,addr,arr,**kw) for t in thetas])
print("current rule, 16dx4a:", mean(rule,cur))
print("current rule, 16dx4a, both dwords of 16 lanes per pass:", mean(rule,cur,both_in_pass=True))
print("transposed S=96:", mean(transp,cur))
for k in (1,3,5,7,9,11,13):
    print("xor k=%d:"%k, mean(xor_sw(k),cur))
print("current rule, 32dx2a:", mean(rule,a32))
print("transposed, 32dx2a:", mean(transp,a32))
