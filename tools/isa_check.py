"""Compiler-dependent properties of the hot kernels, read from the device assembly hipcc emits for csrc/conv.hip
(cross-compiles without a GPU):

    python tools/isa_check.py [substring of a mangled kernel name ...]

per kernel: VGPRs, spills, LDS, waves per SIMD, MFMAs, and every `s_waitcnt vmcnt(N)` BETWEEN the first and the last MFMA
of the kernel -- a vmcnt(0) there drains the software pipeline (the round-2 / round-3 kernels did, once per kernel
offset: the pair-list wait of the offset advance; conv_fwd_flow.h).  tests/test_isa.py pins these numbers."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def device_asm(src='conv.hip', out=None, extra=()):
    out = out or os.path.join('/tmp', 'b2m_%s.s' % src.replace('.', '_'))
    path = os.path.join(ROOT, 'box2mask_amd', 'csrc', src)
    deps = [os.path.join(ROOT, 'box2mask_amd', 'csrc', f) for f in os.listdir(os.path.join(ROOT, 'box2mask_amd', 'csrc'))
            if f.endswith(('.h', '.hip'))]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-munsafe-fp-atomics',
                               '--cuda-device-only', '-S', '-o', out, path] + list(extra), stderr=subprocess.DEVNULL)
    return out


def kernels(asm_path):
    """name -> dict(vgpr, sgpr_spill, vgpr_spill, scratch, lds, occupancy, mfma, loop_waits)"""
    res = {}
    name, body = None, []
    for line in open(asm_path):
        m = re.match(r'^(_Z\w+):', line)
        if m and name is None:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(line)
            if '.end_amdhsa_kernel' in line:
                res[name] = _summarise(body)
                name = None
    # the resource lines follow the kernel body as comments: "; NumVgprs: N" ... collect them in a second pass
    cur = None
    for line in open(asm_path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = m.group(1)
        for key, pat in (('vgpr', r';\s*NumVgprs:\s*(\d+)'), ('agpr', r';\s*NumAgprs:\s*(\d+)'), ('scratch', r';\s*ScratchSize:\s*(\d+)'),
                         ('occupancy', r';\s*Occupancy:\s*(\d+)'), ('lds', r';\s*LDSByteSize:\s*(\d+)'),
                         ('sgpr_spill', r';\s*SGPRSpill.*?:\s*(\d+)'), ('vgpr_spill', r';\s*VGPRSpill.*?:\s*(\d+)')):
            mm = re.match(pat, line.strip())
            if mm and cur in res:
                res[cur][key] = int(mm.group(1))
    return res


def _summarise(body):
    idx = [i for i, l in enumerate(body) if 'v_mfma' in l]
    waits = []
    if idx:
        for l in body[idx[0]:idx[-1] + 1]:
            m = re.search(r's_waitcnt.*vmcnt\((\d+)\)', l)
            if m:
                waits.append(int(m.group(1)))
    return {'mfma': len(idx), 'loop_waits': waits}


if __name__ == '__main__':
    ks = kernels(device_asm())
    pats = sys.argv[1:] or ['conv_fwd_flow_kernel', 'conv_wgrad_flow_kernel', 'conv_stem_kernel', 'conv_1x1']
    for n in sorted(ks):
        if any(p in n for p in pats):
            k = ks[n]
            print('%-62s vgpr %3s spill %s/%s scratch %s lds %5s occ %s mfma %3d  vmcnt in MFMA region: %s' % (
                n, k.get('vgpr'), k.get('sgpr_spill'), k.get('vgpr_spill'), k.get('scratch'), k.get('lds'), k.get('occupancy'),
                k['mfma'], k['loop_waits']))
