"""Executable model of the device's LZ77 + dynamic-Huffman deflate (graphchainer_amd/csrc/hip/gc_deflate.hip: k_deflate_lz_plan / k_deflate_lz_write, GC_GAM_DEVICE_LZ, r6):
the same chunked one-probe match finder, greedy parse, two-queue Huffman lengths and header, statement for statement, in plain Python - zlib's inflate reads what it writes
(python tests/deflate_model.py), and the GPU test holds the kernel's bytes to it (tests/test_gpu_parity.py::test_device_lz_deflate_equals_its_model). Test infrastructure."""
import zlib, sys
LEN_BASE=[3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEN_EXTRA=[0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DIST_BASE=[1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DIST_EXTRA=[0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]
TBITS=12; MINMATCH=4; MAXLEN=258; MAXDIST=32768
def len_sym(L):
    # closed form the kernel uses: L in 3..258
    if L==258: return 28
    v=L-3
    if v<8: return v
    k=v.bit_length()-1      # floor(log2 v) >= 3; k - 2 extra bits
    return 4*(k-1)+((v>>(k-2))&3)
def dist_sym(D):
    v=D-1
    if v<4: return v
    k=v.bit_length()-1
    return 2*k+((v>>(k-1))&1)
for L in range(3,259):
    s=len_sym(L); assert LEN_BASE[s]<=L and (s==28 or L<LEN_BASE[s+1]) , (L,s)
for D in range(1,32769):
    s=dist_sym(D); assert DIST_BASE[s]<=D and (s==29 or D<DIST_BASE[s+1]), (D,s)
def tokens(data):
    n=len(data); table=[0]*(1<<TBITS); cursor=0; out=[]
    for p in range(0,n,64):
        cand=[0]*64; ln=[0]*64; ds=[0]*64
        hs=[None]*64
        for i in range(64):
            pos=p+i
            if pos+4<=n:
                key=data[pos]|data[pos+1]<<8|data[pos+2]<<16|data[pos+3]<<24
                hs[i]=((key*2654435761)&0xffffffff)>>(32-TBITS)
                cand[i]=table[hs[i]]
        for i in range(64):
            if hs[i] is not None: table[hs[i]]=max(table[hs[i]], p+i+1)
        for i in range(64):
            pos=p+i
            if cand[i] and pos-(cand[i]-1)<=MAXDIST:
                c=cand[i]-1; l=0; m=min(MAXLEN,n-pos)
                while l<m and data[c+l]==data[pos+l]: l+=1
                if l>=MINMATCH: ln[i]=l; ds[i]=pos-c
        e=cursor-p
        if e>=64: continue
        i=e
        while i<64 and p+i<n:
            if ln[i]: out.append((ln[i],ds[i])); i+=ln[i]
            else: out.append((0,data[p+i])); i+=1
        cursor=p+i
    return out
def huff_lengths(hist):
    used=sorted((w,a) for a,w in enumerate(hist) if w)
    m=len(used)
    if m==0: return [0]*len(hist), 0
    if m==1:
        l=[0]*len(hist); l[used[0][1]]=1; return l,1
    sw=[w for w,_ in used]; nodeW=[]; leafParent=[0]*m; nodeParent=[0]*m; li=ni=0
    def take(parent):
        nonlocal li,ni
        if li<m and (ni>=len(nodeW) or sw[li]<=nodeW[ni]): leafParent[li]=parent; li+=1; return sw[li-1]
        nodeParent[ni]=parent; ni+=1; return nodeW[ni-1]
    for k in range(m-1):
        nn=len(nodeW); a=take(nn); b=take(nn); nodeW.append(a+b)
    nn=len(nodeW); depth=[0]*nn
    for k in range(nn-2,-1,-1): depth[k]=depth[nodeParent[k]]+1
    lens=[0]*len(hist); deepest=0
    for k in range(m):
        l=depth[leafParent[k]]+1; lens[used[k][1]]=l; deepest=max(deepest,l)
    return lens, deepest
def canon(lens):
    bl=[0]*16
    for l in lens:
        if l: bl[l]+=1
    nxt=[0]*16; c=0
    for l in range(1,16): c=(c+bl[l-1])<<1; nxt[l]=c
    codes=[0]*len(lens)
    for a,l in enumerate(lens):
        if l:
            v=nxt[l]; nxt[l]+=1
            codes[a]=int(format(v,'0%db'%l)[::-1],2)
    return codes
class BitW:
    def __init__(s): s.v=0; s.n=0
    def put(s,val,bits): s.v|=val<<s.n; s.n+=bits
    def bytes(s): return s.v.to_bytes((s.n+7)//8,'little')
def deflate(data):
    toks=tokens(data)
    ll=[0]*286; dd=[0]*30
    for L,x in toks:
        if L: ll[257+len_sym(L)]+=1; dd[dist_sym(x)]+=1
        else: ll[x]+=1
    ll[256]=1
    llen,d1=huff_lengths(ll); dlen,d2=huff_lengths(dd)
    assert d1<=15 and d2<=15
    lc=canon(llen); dc=canon(dlen)
    w=BitW(); w.put(1,1); w.put(2,2); w.put(286-257,5); w.put(30-1,5); w.put(15,4)
    order=[16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15]
    for o in order: w.put(4 if o<16 else 0,3)
    for l in llen+dlen: w.put(int(format(l,'04b')[::-1],2),4)
    for L,x in toks:
        if L:
            s=len_sym(L); w.put(lc[257+s],llen[257+s]); w.put(L-LEN_BASE[s],LEN_EXTRA[s])
            t=dist_sym(x); w.put(dc[t],dlen[t]); w.put(x-DIST_BASE[t],DIST_EXTRA[t])
        else: w.put(lc[x],llen[x])
    w.put(lc[256],llen[256])
    return w.bytes(), toks
if __name__=="__main__":
    import random
    tests=[b"", b"a", b"abcabcabcabcabcabcabcabcabcabcabcabcabc"*40, bytes(random.Random(1).randrange(256) for _ in range(5000)), open(sys.argv[0],'rb').read()*3]
    for t in tests[1:]:
        z,toks=deflate(t)
        back=zlib.decompress(z,-15)
        assert back==t, "mismatch"
        print(len(t), len(z), len(zlib.compress(t,6)), sum(1 for L,_ in toks if L))
    print("model ok")
