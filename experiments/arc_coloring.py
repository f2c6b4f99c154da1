"""How many VGPRs would a hand placement of the planes need?  Live ranges of the two-step loop as circular
arcs, first-fit colouring after cutting at the step boundary (design experiment; result quoted in DESIGN.md)."""
import sys
sys.path.insert(0,'/root/repo/tools')
import gen_lutopt_kernel as G
n,taps=G.load('/root/repo/basebandboard_amd/data/lutopt_256.taps')
order=G.row_order(n,taps)
pos={r:i for i,r in enumerate(order)}
rd=[[] for _ in range(n)]
for r in range(n):
    for c in taps[r]: rd[c].append(pos[r])
first=[min(x) for x in rd]; last=[max(x) for x in rd]; born=[pos[p] for p in range(n)]
def arcs(parked):
    # cycle length 2n: step A positions 0..n-1, step B n..2n-1. For each plane p two versions.
    A=[]  # (start,end,label) with end possibly > 2n meaning wrap
    for p in range(n):
        for s in (0,1):
            base=s*n
            if p in parked:
                # temp copy resident in the READING step: version born in step s is read in step s+1
                A.append((base+n+first[p], base+n+last[p], ('t',p,s)))
            else:
                A.append((base+born[p], base+n+last[p], ('v',p,s)))
    return A
def color(parked):
    A=arcs(parked); T=2*n
    # normalise: start in [0,T), length
    items=[]
    for s,e,l in A:
        ln=e-s; s%=T
        items.append((s,ln,l))
    # load
    load=[0]*T
    for s,ln,l in items:
        for t in range(s,s+ln+1): load[t%T]+=1
    cut=min(range(T),key=lambda t:load[t])
    # rotate so cut at 0
    items=[((s-cut)%T,ln,l) for s,ln,l in items]
    crossing=[it for it in items if it[0]+it[1]>=T]
    non=[it for it in items if it[0]+it[1]<T]
    # colors: each crossing arc gets its own color; occupied intervals on linear timeline [0,T): [0,end-T] and [start,T)
    occ=[]  # per color list of (a,b) busy intervals
    for s,ln,l in crossing:
        occ.append([(0,s+ln-T),(s,T-1)])
    def fits(c,a,b):
        return all(b<x or a>y for x,y in occ[c])
    non.sort(key=lambda it:(it[0],-it[1]))
    for s,ln,l in non:
        a,b=s,s+ln
        best=None
        for c in range(len(occ)):
            if fits(c,a,b):
                best=c;break
        if best is None:
            occ.append([]);best=len(occ)-1
        occ[best].append((a,b))
    return max(load),len(crossing),len(occ)
for B in (180,200,215,225,235):
    parked=G.parking_set(n,taps,order,B)
    ml,cr,ncol=color(parked)
    print('budget',B,'parked',len(parked),'max load',ml,'crossing',cr,'colors',ncol)
