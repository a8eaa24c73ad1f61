import sys, collections, itertools
W,H=3840,2160
NS=(W&~15)//16; NM=(H&~15)//16
TC=(NS+7)//8; TR=(NM+3)//4
pitch=W*3
def tile_lines(tr,tc):
    """lines (128 B) the tile's waves fetch: luma region, chroma sources"""
    L=[]
    x0=tc*128; y0=tr*64
    for y in range(y0,min(y0+64,NM*16)):
        b0=y*pitch+x0*3; b1=b0+min(128,W-x0)*3
        L.extend(range(b0//128,(b1-1)//128+1))
    C=[]
    # chroma: MB (x,y): rows (y/2+i)//... plane offset (y/2+i)*(W/2)+x/2+j, per component byte*3
    for m in range(4):
        mb=tr*4+m
        if mb>=NM: continue
        for i in range(8):
            lin=(mb*8+i)*(W//2)+x0//2      # pixel index in the full-res plane
            b0=lin*3; b1=b0+min(64,(W-x0)//2)*3
            C.extend(range(b0//128,(b1-1)//128+1))
    return L,C
def simulate(order,cap_lines,window=1):
    cache=collections.OrderedDict(); miss=0; total=0
    # interleave accesses of `window` consecutive tiles (concurrency)
    for w0 in range(0,len(order),window):
        acc=[]
        for (tr,tc) in order[w0:w0+window]:
            L,C=tile_lines(tr,tc); acc.append(L+C)
        for chunk in itertools.zip_longest(*acc):
            for ln in chunk:
                if ln is None: continue
                total+=1
                if ln in cache: cache.move_to_end(ln)
                else:
                    miss+=1; cache[ln]=1
                    if len(cache)>cap_lines: cache.popitem(last=False)
    return miss,total
def dfs_rows():
    order=[]; st=[0]
    while st:
        r=st.pop(); order.append(r)
        for c in range(4*r+3,4*r-1,-1):
            if 0<c<TR: st.append(c)
    return [(r,c) for r in order for c in range(TC)]
def natural(): return [(r,c) for r in range(TR) for c in range(TC)]
uniq=W*H*3//128
for name,o in (("natural",natural()),("dfs rows",dfs_rows())):
    for cap,win in ((32768,1),(24000,32),(20000,192)):
        m,t=simulate(o,cap,win); print(f"{name:12s} cap {cap} window {win}: misses {m} = {m/uniq:.4f} x unique lines")
print("--- calibration")
for cap,win in ((32768,8),(30000,8),(28000,16),(26000,16)):
    m,t=simulate(dfs_rows(),cap,win); print(f"dfs rows cap {cap} window {win}: {m/uniq:.4f}")
# 2-D orders.  Source tile of (R,k): (R//4, k//2) left and (R//4, HC + k//2) right, HC = half of the tile columns
HC=TC//2
def order_pairs():
    """walk source PAIRS: pair (R', k') = tiles (R',k') and (R',HC+k'); after a pair, its 8 dependents (4R'..4R'+3, 2k',2k'+1), recursively"""
    done=set(); order=[]
    def visit(R,k):
        if R>=TR or k>=TC or (R,k) in done: return
        done.add((R,k)); order.append((R,k))
    def rec(Rp,kp):
        # the pair
        visit(Rp,kp); visit(Rp,HC+kp)
        for R in range(4*Rp,4*Rp+4):
            for k in (2*kp,2*kp+1):
                if (R,k)!=(Rp,kp) and (R,k)!=(Rp,HC+kp) and R<TR and k<TC and (R,k) not in done:
                    pass
        # dependents are themselves halves of pairs: (R, k) with k<HC pairs with (R, HC+k)
        for R in range(4*Rp,4*Rp+4):
            for k in (2*kp,2*kp+1):
                if R>=TR or k>=TC: continue
                kk = k if k<HC else k-HC
                if (R,kk) not in done: rec(R,kk)
                else:
                    visit(R,k)
    for kp in range(HC): rec(0,kp)
    for R in range(TR):
        for k in range(TC): visit(R,k)
    return order
def order_colblock(bw):
    """DFS over rows, but within column blocks of bw source columns: for block b: columns of sources k' in block, dependents columns 2k'.. """
    order=[];done=set()
    rows=[r for (r,c) in dfs_rows()[::TC]]
    return order
o=order_pairs()
assert len(o)==TR*TC and len(set(o))==TR*TC
for cap,win in ((32768,1),(30000,8),(28000,16)):
    m,t=simulate(o,cap,win); print(f"pairs 2-D dfs cap {cap} window {win}: {m/uniq:.4f}")
print("--- parent-in-the-middle")
def middle_rows(tr_n):
    def children(r): return [c for c in range(4*r,4*r+4) if 0<c<tr_n]
    size={}
    def sz(r):
        size[r]=1+sum(sz(c) for c in children(r)); return size[r]
    sz(0)
    def arr(r):
        ch=sorted(children(r),key=lambda c:size[c])      # smallest first
        left,right=[],[]
        # smallest two adjacent to r (one on each side), the larger ones outside
        for i,c in enumerate(ch):
            a=arr(c)
            if i%2==0: left=a+left if i>=2 else left+a   # i=0 adjacent left; i=2 outside left
            else: right=right+a if i>=3 else a+right
        return left+[r]+right
    return arr(0)
for (w,h) in ((3840,2160),(1920,1080)):
    W,H=w,h; NS=(W&~15)//16; NM=(H&~15)//16; TC=(NS+7)//8; TR=(NM+3)//4; pitch=W*3; uniq=W*H*3//128
    rows=middle_rows(TR); print(w,h,rows)
    o=[(r,c) for r in rows for c in range(TC)]
    for cap,win in ((32768,1),(30000,8),(28000,16),(26000,16)):
        m,t=simulate(o,cap,win); m2,_=simulate(dfs_rows(),cap,win); print(f"  cap {cap} window {win}: middle {m/uniq:.4f}   dfs {m2/uniq:.4f}")
