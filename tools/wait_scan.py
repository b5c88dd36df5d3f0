import re,sys
# compiler-generated vmcnt waits (outside #ASMSTART..#ASMEND) inside loops, per kernel
path=sys.argv[1]; pat=sys.argv[2] if len(sys.argv)>2 else ''
name=None; lines=[]
def report(name, lines):
    inasm=False; tagged=[]
    for l in lines:
        if '#ASMSTART' in l: inasm=True
        tagged.append((l,inasm))
        if '#ASMEND' in l: inasm=False
    labels={}
    for i,(l,_) in enumerate(tagged):
        m=re.match(r'^(\.LBB\w+):',l)
        if m: labels[m.group(1)]=i
    loops=[]
    for i,(l,_) in enumerate(tagged):
        m=re.search(r's_cbranch_\w+\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)',l)
        if m:
            t=m.group(1) or m.group(2)
            if t in labels and labels[t]<i: loops.append((labels[t],i,t))
    # innermost-ish: report loops with mfma
    out=[]
    for a,b,t in loops:
        body=tagged[a:b]
        nm=sum('v_mfma' in x for x,_ in body)
        if nm==0: continue
        cw=[re.search(r'vmcnt\((\d+)\)',x).group(1) for x,ia in body if 'vmcnt' in x and not ia]
        aw=[re.search(r'vmcnt\((\d+)\)',x).group(1) for x,ia in body if 'vmcnt' in x and ia]
        out.append('   loop %s len %d mfma %d | compiler vmcnt: %s | asm vmcnt: %s'%(t,b-a,nm,' '.join(cw) or '-',' '.join(aw[:12]) or '-'))
    if out: print(name[:120]); print('\n'.join(out))
for l in open(path):
    m=re.match(r'^(\w+):\s+; @',l)
    if m: name=m.group(1); lines=[]; continue
    if name is None: continue
    if l.strip().startswith('.Lfunc_end'):
        if re.search(pat,name): report(name,lines)
        name=None; continue
    lines.append(l)
