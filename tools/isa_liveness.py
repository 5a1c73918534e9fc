"""VGPR liveness of a step kernel from its ISA: which registers are live where, and what stays live across the whole kernel.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -gline-tables-only --cuda-device-only -S rsr_mjx_amd/csrc/rsr_mjx.hip -o dev.s [-DRSR_WAVES_PER_EU=3]
    python tools/isa_liveness.py dev.s [Li22 = cube | Li15 = T-shape]
Backward data flow over the kernel's basic blocks on the final instruction stream (defs / uses of v registers parsed from the text;
read-modify-write forms: v_fmac, v_writelane, partial-row DPP).  Prints the maximum, the source lines at the maximum, the live
count along the program, and the registers live at every probe point with their defining instructions.  Used in round 3 to see
why a 168-VGPR build spills 105 registers: the peak is ls_eval inside the line-search loop, ~50 of the 73 registers
live across the whole 256-register kernel are hoisted constants, and nothing but the line search is above 153 in the 168 build."""
import re,sys,collections
path=sys.argv[1]; pat=sys.argv[2] if len(sys.argv)>2 else 'Li22'
s=open(path).read().split('\n')
start=[i for i,l in enumerate(s) if l.startswith('_ZN3rsr11step_kernelINS_4DimsI'+pat)][0]
end=start
while not s[end].startswith('.Lfunc_end'): end+=1
files={}
for l in s:
    m=re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?',l)
    if m: files[int(m.group(1))]=(m.group(3) or m.group(2)).split('/')[-1]
def regs(tok):
    out=set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b',tok):
        if m.group(1) is not None: out.update(range(int(m.group(1)),int(m.group(2))+1))
        else: out.add(int(m.group(3)))
    return out
NODEF=('ds_write','global_store','scratch_store','buffer_store','flat_store','v_cmp','s_','v_readlane','v_readfirstlane','global_atomic','ds_add','ds_max','ds_min','ds_or','ds_and','ds_bpermute_NEVER','v_nop','buffer_wbl2','buffer_inv','ds_nop')
RMW=('v_fmac','v_writelane','v_mac','v_pk_fmac','v_dot2c')
ins=[]  # (op, defs, uses, loc, text)
labels={}
cur=None
for l in s[start+1:end]:
    m=re.match(r'\s*\.loc\s+(\d+)\s+(\d+)',l)
    if m: cur=(files.get(int(m.group(1)),'?'),int(m.group(2))); continue
    t=l.split(';')[0].strip()
    if not t or t.startswith('.') and not t.endswith(':'): continue
    if t.endswith(':'):
        labels[t[:-1]]=len(ins); continue
    parts=t.split(None,1)
    op=parts[0]; rest=parts[1] if len(parts)>1 else ''
    ops=[x.strip() for x in rest.split(',')]
    defs=set(); uses=set()
    if op.startswith(NODEF) and not (op.startswith('global_atomic') and 'sc0' in rest) and not op.endswith('_rtn_f32'):
        for o in ops: uses|=regs(o)
    else:
        if ops:
            defs=regs(ops[0])
            for o in ops[1:]: uses|=regs(o)
            if op.startswith(RMW) or ('dpp' in op and ('bound_ctrl' not in rest or ('row_mask:0xf' not in rest and 'row_mask' in rest))): uses|=defs
            # partial-lane dpp writes keep old value
            if 'row_mask:0xa' in rest or 'row_mask:0xc' in rest: uses|=defs
    ins.append((op,defs,uses,cur,t))
n=len(ins)
succ=[[] for _ in range(n)]
for i,(op,d,u,loc,t) in enumerate(ins):
    if op=='s_endpgm': continue
    if op=='s_branch':
        succ[i].append(labels[t.split()[1]]); continue
    if op.startswith('s_cbranch'):
        succ[i].append(labels[t.split()[1]])
    if i+1<n: succ[i].append(i+1)
live_in=[set() for _ in range(n)]
changed=True; it=0
while changed:
    changed=False; it+=1
    for i in range(n-1,-1,-1):
        out=set()
        for j in succ[i]: out|=live_in[j]
        new=(out-ins[i][1])|ins[i][2]
        if new!=live_in[i]: live_in[i]=new; changed=True
print('instructions',n,'iterations',it,'max live',max(len(x) for x in live_in))
byline=collections.defaultdict(int)
for i in range(n):
    k=ins[i][3]; byline[k]=max(byline[k],len(live_in[i]))
# coarse: per (file, line//1) top
for k,v in sorted(byline.items(), key=lambda x:-x[1])[:40]: print(k,v)
# profile along program order, sampled
print('--- program order (every 250 instr): idx line live')
for i in range(0,n,250): print(i, ins[i][3], len(live_in[i]))
# registers live at all of a set of probe points
probes=[int(n*f) for f in (0.35,0.55,0.70,0.84,0.90,0.95)]          # (positions along the program: collision .. solver .. integrate)
common=set.intersection(*[live_in[p] for p in probes])
print('common live across probes',len(common),sorted(common))
# find defs of those registers (all def sites)
defsites=collections.defaultdict(list)
for i,(op,d,u,loc,t) in enumerate(ins):
    for r in d:
        if r in common: defsites[r].append((i,loc,t[:90]))
for r in sorted(common):
    ds=defsites[r]
    print('v%d'%r, len(ds), [ (i,loc) for i,loc,t in ds[:4]], ds[0][2] if ds else '')
