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
        base = [HIPCC, '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-munsafe-fp-atomics', '--cuda-device-only']
        # `-S` PRINTS an asm statement whatever registers hipcc substituted for its operands; only the assembler checks them (round 6:
        # a VGPR pair for an "s" operand went through every test here while the library no longer built).  So the text is assembled too.
        r = subprocess.run(base + ['-c', '-o', os.devnull, path] + list(extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            if os.path.exists(out):
                os.remove(out)
            raise RuntimeError('hipcc does not assemble %s:\n%s' % (src, r.stdout.decode()[-3000:]))
        subprocess.check_call(base + ['-S', '-o', out, path] + list(extra), stderr=subprocess.DEVNULL)
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


def kernel_body(asm_path, name):
    s = open(asm_path).read()
    i = s.index(name + ':')
    return s[i:s.index('.end_amdhsa_kernel', i)].split('\n')


_REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def _regs(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def inflight_violations(body):
    """Hand-issued loads (global_load_* inside an asm statement) are invisible to hipcc's wait insertion: until the counted
    `s_waitcnt vmcnt(N)` that covers a load has run, NOTHING may read or write the registers it is going to fill -- not the
    MFMAs (a wrong count), not a compiler-generated copy, not another value the allocator put there.  This traces the kernel
    with the in-order load queue vmcnt counts against (every global_load enters it, `vmcnt(N)` leaves the N youngest) and
    returns the instructions that touch a register of a hand-issued load still in the queue.  The trace: unconditional
    branches are followed, conditional ones fall through unless they jump BACK (a loop's back edge: taken until the target
    has been passed twice, so values carried around a loop are followed once); code the trace never reaches is traced
    afterwards, block by block, from an empty queue.  A second trace takes every forward branch that jumps over a counted
    wait (and no load): the consumer of a register is skipped (a k-step without pairs, an absent row group), its load is
    still in flight behind the branch -- and the register, dead in hipcc's eyes, is a favourite for the next temporary."""
    labels = {}
    for k, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = k
    bad = {}

    def skips_a_wait(k, t):
        region = body[k:t]
        return (any(re.search(r's_waitcnt.*vmcnt', l) and 'ASMSTART' in region[i - 1] for i, l in enumerate(region) if i)
                and not any('global_load' in l or 'buffer_load' in l for l in region))

    def trace(start, visits, skip):
        queue = []          # (hand_issued, destination registers), oldest first
        in_asm = False
        k = start
        while k < len(body):
            if visits[k] >= 2:
                return
            visits[k] += 1
            raw = body[k].strip()
            k += 1
            if raw.startswith(';;#ASMSTART'):
                in_asm = True
                continue
            if raw.startswith(';;#ASMEND'):
                in_asm = False
                continue
            l = raw.split(';')[0].strip()
            if not l or l.endswith(':') or l.startswith('.'):
                continue
            if l.startswith('s_endpgm'):
                return
            m = re.match(r's_waitcnt.*vmcnt\((\d+)\)', l)
            if m:
                n = int(m.group(1))
                if len(queue) > n:
                    queue = queue[len(queue) - n:] if n else []
                continue
            m = re.match(r's_(c?)branch\w*\s+(\.LBB\d+_\d+)', l)
            if m:
                t = labels.get(m.group(2))
                if t is None:
                    continue
                if not m.group(1):                       # unconditional
                    k = t
                elif t < k and visits[t] < 2:            # back edge
                    k = t
                elif t > k and skip and skips_a_wait(k, t):
                    k = t
                continue
            ops = l.split(None, 1)
            regs = _regs(ops[1]) if len(ops) > 1 else set()
            pending = set()
            for hand, dst in queue:
                if hand:
                    pending |= dst
            hit = regs & pending
            if ops[0].startswith(('global_load', 'buffer_load', 'flat_load')):
                dst = _regs(ops[1].split(',')[0])
                hit = (regs - dst) & pending          # (a second load into a register in flight lands in order: harmless)
                queue.append((in_asm, dst))
            if hit and k - 1 not in bad:
                bad[k - 1] = (k - 1, l, sorted(hit))

    for skip in (False, True):
        visits = [0] * len(body)
        trace(0, visits, skip)
        for k in range(len(body)):
            if visits[k] == 0 and re.match(r'^\.LBB\d+_\d+:', body[k]):
                trace(k, visits, skip)
    return [bad[k] for k in sorted(bad)]


def mfma_operand_violations(body, states=2):
    """An MFMA written as an asm statement is invisible to hipcc's hazard recogniser: a VGPR a VALU instruction wrote fewer than
    `states` wait states earlier is read OLD by the MFMA's A / B operand (round 6: the half weight gradient's 2 x 2 blocks took
    bz[0] unmasked in the first MFMA of every k-step).  Linear scan of the text, branches not followed: every instruction is one
    state, `s_nop N` is N + 1; a label keeps the history (the fall-through path is a real path).  Returns (line, mfma text,
    registers, states since the write)."""
    bad = []
    recent = []          # (destination registers, states since) of VALU writes, youngest last
    for k, raw in enumerate(body):
        l = raw.split(';')[0].strip()
        if not l or l.endswith(':') or l.startswith('.') or l.startswith(';;'):
            continue
        ops = l.split(None, 1)
        op = ops[0]
        cost = 1
        m = re.match(r's_nop\s+(\d+)', l)
        if m:
            cost = int(m.group(1)) + 1
        if op.startswith('v_mfma') and len(ops) > 1:
            parts = [x.strip() for x in ops[1].split(',')]
            # "v[12:15], v26, v29, v[12:15]": operands 1 and 2 are A and B
            ab = _regs(parts[1]) | _regs(parts[2]) if len(parts) >= 3 else set()
            for dst, age in recent:
                if age < states and dst & ab:
                    bad.append((k, l, sorted(dst & ab), age))
        recent = [(d, a + cost) for d, a in recent if a + cost < 8]
        if op.startswith('v_') and not op.startswith(('v_mfma', 'v_cmp', 'v_nop')) and len(ops) > 1:
            recent.append((_regs(ops[1].split(',')[0]), 0))
    return bad


if __name__ == '__main__':
    ks = kernels(device_asm())
    pats = sys.argv[1:] or ['conv_fwd_flow_kernel', 'conv_wgrad_flow_kernel', 'conv_stem_kernel', 'conv_1x1']
    for n in sorted(ks):
        if any(p in n for p in pats):
            k = ks[n]
            print('%-62s vgpr %3s spill %s/%s scratch %s lds %5s occ %s mfma %3d  vmcnt in MFMA region: %s' % (
                n, k.get('vgpr'), k.get('sgpr_spill'), k.get('vgpr_spill'), k.get('scratch'), k.get('lds'), k.get('occupancy'),
                k['mfma'], k['loop_waits']))
            for v in inflight_violations(kernel_body(device_asm(), n)):
                print('    register of a load in flight touched: line %d  %s  %s' % v)
            for v in mfma_operand_violations(kernel_body(device_asm(), n)):
                print('    MFMA operand written by a VALU instruction %d state(s) earlier: line %d  %s  %s' % (v[3], v[0], v[1], v[2]))
