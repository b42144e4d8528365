"""Asm-level variants of the old role-1 kernel (K = 8, linear tails) for tools/probe/gw_fault_hsaco.sh: the compiler's own assembly
(/tmp/probe_tails.s, written by tools/probe/build_old_bwd_variants.sh) with ONE class of waits strengthened or inserted inside that
kernel, re-assembled to tools/probe/build/gw_<tag>.hsaco.  DESIGN.md section 4d lists what each variant showed."""
import re, subprocess, sys, os
src=open('/tmp/probe_tails.s').read().split('\n')
name='_ZN2fc24rq_fused_backward_kernelILi8ELb1ELi1EEEvNS_8RQParamsEfNS_7BwdArgsE'
start=[i for i,l in enumerate(src) if l.startswith(name+':')][0]
end=[i for i in range(start,len(src)) if src[i].strip().startswith('s_endpgm')][0]
def build(tag, fn):
    out=list(src)
    body=[]
    for l in src[start:end+1]:
        body+=fn(l)
    out[start:end+1]=body
    os.makedirs('/tmp/variants',exist_ok=True)
    p='/tmp/variants/%s.s'%tag
    open(p,'w').write('\n'.join(out))
    subprocess.run(['/opt/rocm/lib/llvm/bin/clang','-x','assembler','-target','amdgcn-amd-amdhsa','-mcpu=gfx950','-c',p,'-o','/tmp/variants/%s.o'%tag],check=True)
    subprocess.run(['/opt/rocm/lib/llvm/bin/ld.lld','-shared','/tmp/variants/%s.o'%tag,'-o',os.path.join(os.path.dirname(os.path.abspath(__file__)),'build','gw_%s.hsaco'%tag)],check=True)
    print(tag, 'lines', len(body))
ident=lambda l:[l]
def vm0(l):
    return [re.sub(r'vmcnt\(\d+\)','vmcnt(0)',l)] if l.strip().startswith('s_waitcnt') else [l]
def lgkm0(l):
    return [re.sub(r'lgkmcnt\(\d+\)','lgkmcnt(0)',l)] if l.strip().startswith('s_waitcnt') else [l]
def after(prefixes, wait):
    def f(l):
        t=l.strip()
        return [l, '\t'+wait] if any(t.startswith(p) for p in prefixes) else [l]
    return f
build('base', ident)
build('vm0', vm0)
build('lgkm0', lgkm0)
build('scratchload_wait', after(['scratch_load'], 's_waitcnt vmcnt(0)'))
build('scratchstore_wait', after(['scratch_store'], 's_waitcnt vmcnt(0)'))
build('dswrite_wait', after(['ds_write'], 's_waitcnt lgkmcnt(0)'))
build('globalload_wait', after(['global_load'], 's_waitcnt vmcnt(0)'))

# ---- second batch: is it the wave-private strip (ds_write_b16 -> ds_read_b128 of the same wave), or just delay? ----------
def off(l):
    m=re.search(r'offset:(\d+)',l)
    return int(m.group(1)) if m else 0
def strip_wait(l):
    t=l.strip()
    return [l,'\ts_waitcnt lgkmcnt(0)'] if t.startswith('ds_write_b16') and off(t)>=45056 else [l]
def strip_read_fence(l):
    t=l.strip()
    return ['\ts_waitcnt lgkmcnt(0)',l] if t.startswith('ds_read_b128') and off(t)>=45056 else [l]
def nonstrip_wait(l):
    t=l.strip()
    return [l,'\ts_waitcnt lgkmcnt(0)'] if t.startswith('ds_write') and not (t.startswith('ds_write_b16') and off(t)>=45056) else [l]
state={'n':0}
def delay_only(l):
    if 's_barrier' in l:
        state['n']+=1
        if state['n']==1:
            return [l]+['\ts_sleep 127']*16
    return [l]
build('strip_wait', strip_wait)
build('strip_read_fence', strip_read_fence)
build('nonstrip_wait', nonstrip_wait)
build('delay_only', delay_only)
