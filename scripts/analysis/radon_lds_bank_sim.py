"""Simulation of the Radon kernel's LDS bank pattern: conflict factor of the half-wave's ds_read2_b32
row fetches for row strides and lane arrangements.  python scripts/analysis/radon_lds_bank_sim.py"""
import numpy as np
rng=np.random.default_rng(0)
n_alpha=n_t=768; W=H=1024; D=np.sqrt(2)*1024
def cycles(words):  # words: array (lanes, k) of dword indices for one pass; 32 banks
    tot=0
    b=words%32
    c=0
    for bank in range(32):
        u=np.unique(words[b==bank])
        c=max(c,len(u))
    return c
def sim(S_rule, lane_perm=None, trials=400, passes=32):
    tot=0; ideal=0
    for _ in range(trials):
        ia0=rng.integers(0,n_alpha//16)*16; it0=rng.integers(8,n_t//16-8)*16
        w=rng.integers(0,4)
        a_idx=ia0+4*w+np.repeat(np.arange(4),16); t_idx=it0+np.tile(np.arange(16),4)
        if lane_perm is not None:
            a_idx,t_idx=lane_perm(ia0,it0,w)
        alpha=(a_idx/n_alpha-0.5)*np.pi; tau=(t_idx/n_t-0.5)*D
        l0,l1=-np.sin(alpha),np.cos(alpha)   # normal
        l2=-tau - 0.5*W*l0 - 0.5*H*l1
        ox,oy=-l2*l0,-l2*l1; dx,dy=l1,-l0
        # along-line coordinate: common U plus per-lane jitter in [0,0.66)
        U=rng.uniform(-300,300)
        # param t such that point = o + t d; choose t = U' + jitter where U' aligns lanes: use projection onto mean direction
        jit=rng.uniform(0,0.66,64)
        t=U+jit
        for i in range(8):
            x=ox+(t+0.66*i)*dx+0.5; y=oy+(t+0.66*i)*dy+0.5
            fi=np.floor(x-0.5).astype(int); fj=np.floor(y-0.5).astype(int)
            nx,ny=l0.mean(),l1.mean()
            S=S_rule(nx,ny)
            idx=fj*S+fi+100000
            for base in (idx, idx+S):
                words=np.stack([base,base+1],1)
                for h in range(0,64,passes):
                    tot+=cycles(words[h:h+passes]); ideal+=2*passes//32
    return tot/ideal
rule9795=lambda nx,ny: 97 if abs(nx+ny)>=abs(nx-ny) else 95
print("current 97/95 rule, 32-lane passes:", sim(rule9795))
print("stride 97 only:", sim(lambda nx,ny:97))
print("stride 96:", sim(lambda nx,ny:96))
for S in (99,101,103,105,107,109,111,113):
    print("stride",S, sim(lambda nx,ny,S=S:S, trials=200))
print("---- per-direction best stride")
def sim_dir(theta, S, trials=60):
    tot=0; ideal=0
    for _ in range(trials):
        ia0=int((theta/np.pi+0.5)*n_alpha)//16*16; it0=rng.integers(8,n_t//16-8)*16
        w=rng.integers(0,4)
        a_idx=ia0+4*w+np.repeat(np.arange(4),16); t_idx=it0+np.tile(np.arange(16),4)
        alpha=(a_idx/n_alpha-0.5)*np.pi; tau=(t_idx/n_t-0.5)*D
        l0,l1=-np.sin(alpha),np.cos(alpha)
        l2=-tau - 0.5*W*l0 - 0.5*H*l1
        ox,oy=-l2*l0,-l2*l1; dx,dy=l1,-l0
        U=rng.uniform(-300,300); t=U+rng.uniform(0,0.66,64)
        for i in range(6):
            x=ox+(t+0.66*i)*dx+0.5; y=oy+(t+0.66*i)*dy+0.5
            fi=np.floor(x-0.5).astype(int); fj=np.floor(y-0.5).astype(int)
            idx=fj*S+fi+100000
            for base in (idx, idx+S):
                words=np.stack([base,base+1],1)
                for h in (0,32):
                    tot+=cycles(words[h:h+32]); ideal+=2
    return tot/ideal
cands=[91,93,95,97,99,101,103]
best=[];cur=[]
for th in np.linspace(-np.pi/2+0.01,np.pi/2-0.01,24):
    r={S:sim_dir(th,S) for S in cands}
    nx,ny=-np.sin(th),np.cos(th)
    c=r[rule9795(nx,ny)]
    b=min(r,key=r.get)
    best.append(r[b]);cur.append(c)
    print("theta %6.1f deg: current %.2f  best S=%d %.2f   all: %s"%(np.degrees(th),c,b,r[b]," ".join("%d:%.2f"%(S,r[S]) for S in cands)))
print("mean current %.3f, mean best %.3f"%(np.mean(cur),np.mean(best)))
print("---- lane arrangements (best of strides 95/97 rule)")
def sim_arr(theta, S, arr, trials=60):
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
        for i in range(6):
            x=ox+(t+0.66*i)*dx+0.5; y=oy+(t+0.66*i)*dy+0.5
            fi=np.floor(x-0.5).astype(int); fj=np.floor(y-0.5).astype(int)
            idx=fj*S+fi+100000
            for base in (idx, idx+S):
                words=np.stack([base,base+1],1)
                for h in (0,32):
                    tot+=cycles(words[h:h+32]); ideal+=2
    return tot/ideal
arrs={
 "16d x 4a (current)": lambda ia0,it0,w:(ia0+4*w+np.repeat(np.arange(4),16), it0+np.tile(np.arange(16),4)),
 "32d x 2a": lambda ia0,it0,w:(ia0+2*w+np.repeat(np.arange(2),32), it0+np.tile(np.arange(32),2)),
 "64d x 1a": lambda ia0,it0,w:(ia0+w+np.zeros(64,int), it0+np.arange(64)),
 "16d x (a,a+8) pairs": lambda ia0,it0,w:(ia0+np.repeat(np.array([2*w,2*w+8,2*w+1,2*w+9]),16), it0+np.tile(np.arange(16),4)),
 "8d x 8a": lambda ia0,it0,w:(ia0+8*(w%2)+np.repeat(np.arange(8),8), it0+8*(w//2)+np.tile(np.arange(8),8)),
}
for name,arr in arrs.items():
    vals=[]
    for th in np.linspace(-np.pi/2+0.01,np.pi/2-0.01,16):
        nx,ny=-np.sin(th),np.cos(th)
        vals.append(min(sim_arr(th,S,arr,30) for S in (95,97)))
    print("%-24s mean conflict factor %.3f (min %.2f max %.2f)"%(name,np.mean(vals),min(vals),max(vals)))
