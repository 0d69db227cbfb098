"""Load/wait token scan of the compiled kernels: for each kernel of a .hip file prints, per basic block, the
sequence of global loads (L), stores (S), atomics (A) and `s_waitcnt vmcnt(n)` (wn).  A block that reads
`L w0 L w0` is a chain of serialized memory round trips; `L L L w2 w1 w0` is one round trip.

    python tools/isa_loads.py rpe.hip [kernel-substring ...]
"""
import os, re, subprocess, sys

def main():
    src = sys.argv[1]; want = sys.argv[2:]
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stratified_transformer_amd', 'csrc')
    out = '/tmp/isa'; os.makedirs(out, exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
                           '-save-temps', '-Wno-pass-failed', '-c', os.path.join(csrc, src), '-o', os.path.join(out, 'x.o')], cwd=out)
    asm = open(os.path.join(out, src.replace('.hip', '') + '-hip-amdgcn-amd-amdhsa-gfx950.s')).read().split('\n')
    name, toks = None, []
    def flush():
        if not name: return
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        if not want or any(w in dem for w in want):
            print(dem[:110]); print('   ', ' '.join(toks))
    for ln in asm:
        m = re.match(r'^(_Z\w+):', ln)
        if m:
            flush(); name, toks = m.group(1), []
            continue
        s = ln.strip()
        if re.match(r'^\.LBB\d+_\d+:', s): toks.append('|')
        elif s.startswith('global_load') or s.startswith('buffer_load'): toks.append('L')
        elif s.startswith('global_store'): toks.append('S')
        elif s.startswith('global_atomic'): toks.append('A')
        elif s.startswith('s_waitcnt'):
            v = re.search(r'vmcnt\((\d+)\)', s)
            if v: toks.append('w' + v.group(1))
        elif s.startswith('s_endpgm'):
            flush(); name = None
    
main()
