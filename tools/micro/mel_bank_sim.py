"""Diagnostic (CPU only): LDS bank-conflict model of the mel projection's power-chunk reads (asrk_logmel.hip) for the list assignment of
asrk_logmel_plan_create, and a simulated-annealing search over assignments / orders / half-swap bits.
    python tools/micro/mel_bank_sim.py [anneal iterations]"""
import sys, itertools, random
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from asr_amd.mel import slaney_mel_filter_bank
NB=201; LANES=20; PPAIR=424
GROUPS=[list(range(0,4))+list(range(12,16))+list(range(20,28)),
        list(range(4,12))+list(range(16,20))+list(range(28,32))]
GROUPS=GROUPS+[[l+32 for l in g] for g in GROUPS]

def filters(n_mels):
    fb=slaney_mel_filter_bank(n_mels).astype(np.float32)  # [201][n_mels]?
    if fb.shape[0]!=NB: fb=fb.T
    out=[]
    for m in range(n_mels):
        nz=np.nonzero(fb[:,m])[0]
        lo,hi=(nz[0],nz[-1]) if len(nz) else (0,0)
        out.append((lo//4, hi//4-lo//4+1))
    return out

def baseline_lists(flt):
    order=sorted(range(len(flt)), key=lambda m:-flt[m][1])  # stable
    lists=[[] for _ in range(LANES)]; per=[0]*LANES
    for m in order:
        q=min(range(LANES), key=lambda l:(per[l],l))
        lists[q].append(m); per[q]+=flt[m][1]
    return lists

def steps_of(lists, flt):
    n=max(sum(flt[m][1] for m in L) for L in lists); n=(n+1)//2*2
    tab=np.zeros((n,LANES),dtype=int)  # chunk index per step/list (padding: chunk 0)
    for q,L in enumerate(lists):
        e=0
        for m in L:
            c0,nch=flt[m]
            for c in range(nch): tab[e,q]=c0+c; e+=1
    return tab

def cost(tab, hswf=lambda j:(j&1)<<4):
    tot=0
    for e in range(tab.shape[0]):
        for half in (0,16):
            for G in GROUPS:
                quads={}
                for l in G:
                    if l>=60: continue
                    g,j=divmod(l,20)
                    addr=g*PPAIR*4+tab[e,j]*32+(hswf(j)^half)
                    quads.setdefault((addr//16)%16,set()).add(addr)
                tot+=max(len(v) for v in quads.values()) if quads else 0
    return tot  # LDS passes; ideal = steps*2*4

for n_mels in (80,128):
    flt=filters(n_mels)
    L=baseline_lists(flt); tab=steps_of(L,flt)
    print(n_mels,"steps",tab.shape[0],"cost",cost(tab),"ideal",tab.shape[0]*8)

def anneal(flt, n_mels, iters=20000, seed=1, maxlen=10):
    rnd=random.Random(seed)
    L=[list(x) for x in baseline_lists(flt)]
    hs=[(j&1) for j in range(LANES)]
    def c(L,hs): return cost(steps_of(L,flt), lambda j:hs[j]<<4)
    cur=c(L,hs); best=cur; bestL=[list(x) for x in L]; besth=list(hs)
    T=2.0
    for it in range(iters):
        L2=[list(x) for x in L]; h2=list(hs)
        r=rnd.random()
        if r<0.15:
            j=rnd.randrange(LANES); h2[j]^=1
        elif r<0.55:
            j=rnd.randrange(LANES)
            if len(L2[j])>1:
                a,b=rnd.sample(range(len(L2[j])),2); L2[j][a],L2[j][b]=L2[j][b],L2[j][a]
        else:
            j,k=rnd.sample(range(LANES),2)
            if L2[j] and L2[k]:
                a=rnd.randrange(len(L2[j])); b=rnd.randrange(len(L2[k]))
                L2[j][a],L2[k][b]=L2[k][b],L2[j][a]
                if max(sum(flt[m][1] for m in L2[j]),sum(flt[m][1] for m in L2[k]))>maxlen: continue
        n=c(L2,h2)
        if n<=cur or rnd.random()<np.exp((cur-n)/T):
            L,hs,cur=L2,h2,n
            if cur<best: best,bestL,besth=cur,[list(x) for x in L],list(hs)
        T=max(0.05,T*0.9997)
    return best,bestL,besth

if __name__=="__main__" and len(sys.argv)>1:
    for n_mels in (80,128):
        flt=filters(n_mels)
        b,L,h=anneal(flt,n_mels,iters=int(sys.argv[1]))
        print(n_mels,"annealed cost",b,"hs",h)
