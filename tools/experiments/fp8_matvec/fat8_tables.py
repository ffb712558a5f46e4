import random, sys
random.seed(int(sys.argv[1]) if len(sys.argv)>1 else 0)
# ops and precedence
ops={}
for i in range(16): ops[f'E{i}']=(8,[])
for p in range(8):
    ops[f'H{p}']=(4,[f'E{2*p}',f'E{2*p+1}'])
    ops[f'M{2*p}']=(4,[f'H{p}'])
    ops[f'M{2*p+1}']=(4,[f'M{2*p}'])
for d in range(4):
    ops[f'A{d}']=(8,[f'H{2*d}',f'H{2*d+1}']); ops[f'B{d}']=(8,[f'A{d}']); ops[f'C{d}']=(4,[f'B{d}'])
names=list(ops)
def deadline(k,S):
    if k[0]=='E': return S-1
    if k[0]=='H': return S-1 if int(k[1:])<4 else S+3
    if k[0]=='M': return S-1 if int(k[1:])<8 else S+4
    return S+9
def solve(S, cap, LAG, iters=200000):
    # LAG: fixed loads that the PREVIOUS chain's lag ops put on this block's slots (same table assumed for both neighbours' lag part)
    # state: slot per op in [0, S+9]
    slot={}
    # init: topological earliest
    for k in names:
        lo=max([slot[d]+1 for d in ops[k][1]],default=0)
        slot[k]=min(lo,deadline(k,S))
    def cost(slot):
        load=[0]*(S+10)
        for k,v in slot.items(): load[v]+=ops[k][0]
        tot=[load[i]+(load[S+i] if S+i<S+10 else 0) for i in range(S)]   # lag part of an identical neighbour chain lands on slots 0..9
        c=0
        for i in range(S):
            over=tot[i]-cap[i]
            if over>0: c+=over*over*4
            c+=(tot[i]/cap[i])**2
        ne=[0]*(S+10)
        for k,v in slot.items():
            if k[0]=='E': ne[v]+=1
        for i in range(S):
            mx=2 if cap[i]<=40 else 6
            if ne[i]>mx: c+=100*(ne[i]-mx)
        return c,tot
    def feasible(k,v,slot):
        if v<0 or v>deadline(k,S): return False
        for d in ops[k][1]:
            if slot[d]>=v: return False
        for k2,(du,deps) in ops.items():
            if k in deps and slot[k2]<=v: return False
        return True
    cur,tot=cost(slot); T=5.0
    best=(cur,dict(slot),tot)
    import math
    for it in range(iters):
        k=random.choice(names); v=slot[k]+random.choice([-2,-1,1,2])
        if not feasible(k,v,slot): continue
        old=slot[k]; slot[k]=v
        c,t=cost(slot)
        if c<cur or random.random()<math.exp((cur-c)/T):
            cur=c
            if c<best[0]: best=(c,dict(slot),t)
        else: slot[k]=old
        T=max(0.01,T*0.99997)
    return best
for S,cap in ((10,[31]*10),(12,[24]*10+[56,56])):
    c,slot,tot=solve(S,cap,None)
    print('S',S,'cost',round(c,2),'loads',tot)
    for fam,n in (('E',16),('H',8),('M',16),('A',4),('B',4),('C',4)):
        print('  ',fam,[slot[f'{fam}{i}'] for i in range(n)])
