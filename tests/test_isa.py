"""Compiler-dependent properties of the hot kernels, pinned on the device assembly hipcc emits for csrc/conv.hip (no GPU
needed: hipcc cross-compiles gfx950).  The pipelines of conv_fwd_flow_kernel / conv_wgrad_flow_kernel / conv_stem_kernel
depend on things a toolchain bump can silently undo -- registers under the occupancy steps (168 / 128 / 88), no scratch, and
counted `s_waitcnt vmcnt(N)` waits inside the MFMA loops (a vmcnt(0) there drains the software pipeline: the round-2 and
round-3 builds had one at every kernel-offset advance, found in round 4 with tools/isa_check.py).  A regression fails HERE,
in the CPU suite and in build(), instead of showing up as 20 % in the benchmark."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


@pytest.fixture(scope='module')
def kernels():
    import isa_check
    if not os.path.exists(isa_check.HIPCC):
        pytest.skip('hipcc not available')
    return isa_check.kernels(isa_check.device_asm())


# mangled name -> (max VGPRs, waves per SIMD, needs counted waits in the MFMA loop)
PINNED = {
    # forward / data gradient, hand-issued loads (the shipped variants): 48- and 32-column strips, un-split and 4 slices per workgroup
    '_Z20conv_fwd_flow_kernelILi2ELi3ELi0ELi1ELi1ELi0ELi0EEv8ConvArgs': (168, 3, True),
    '_Z20conv_fwd_flow_kernelILi2ELi2ELi0ELi1ELi1ELi0ELi0EEv8ConvArgs': (128, 4, True),
    '_Z20conv_fwd_flow_kernelILi2ELi3ELi0ELi4ELi1ELi0ELi0EEv8ConvArgs': (168, 3, True),
    '_Z20conv_fwd_flow_kernelILi2ELi2ELi0ELi4ELi1ELi0ELi0EEv8ConvArgs': (128, 4, True),
    # (round 6) two slices per item, workgroups of two waves (B2M_CONV_SPLIT2, off by default)
    '_Z20conv_fwd_flow_kernelILi2ELi3ELi0ELi2ELi1ELi0ELi0EEv8ConvArgs': (168, 3, True),
    '_Z20conv_fwd_flow_kernelILi2ELi2ELi0ELi2ELi1ELi0ELi0EEv8ConvArgs': (128, 4, True),
    # the half-precision inference variants (b2m_conv_fwd_h): 32-channel chunks with 2 / 3 steps in flight, 16-channel chunks
    '_Z20conv_fwd_flow_kernelILi2ELi3ELi0ELi1ELi1ELi1ELi0EEv8ConvArgs': (168, 3, True),
    '_Z20conv_fwd_flow_kernelILi2ELi2ELi0ELi1ELi1ELi1ELi0EEv8ConvArgs': (128, 4, True),
    '_Z20conv_fwd_flow_kernelILi3ELi3ELi0ELi1ELi1ELi1ELi0EEv8ConvArgs': (168, 3, True),
    '_Z20conv_fwd_flow_kernelILi3ELi2ELi0ELi4ELi1ELi1ELi0EEv8ConvArgs': (168, 3, True),
    '_Z20conv_fwd_flow_kernelILi2ELi2ELi0ELi1ELi1ELi2ELi0EEv8ConvArgs': (128, 4, True),
    '_Z20conv_fwd_flow_kernelILi2ELi3ELi0ELi4ELi1ELi2ELi0EEv8ConvArgs': (168, 3, True),
    # ... in 64-column strips (round 6: output channels in multiples of 64 on maps of 256 tiles and more): two waves per SIMD
    '_Z20conv_fwd_flow_kernelILi2ELi4ELi0ELi1ELi1ELi1ELi0EEv8ConvArgs': (184, 2, True),
    '_Z20conv_fwd_flow_kernelILi2ELi4ELi0ELi4ELi1ELi1ELi0EEv8ConvArgs': (184, 2, True),
    '_Z20conv_fwd_flow_kernelILi3ELi4ELi0ELi1ELi1ELi1ELi0EEv8ConvArgs': (216, 2, True),
    '_Z20conv_fwd_flow_kernelILi2ELi4ELi0ELi1ELi1ELi2ELi0EEv8ConvArgs': (152, 2, True),
    # the transposed k2s2 maps in scatter form (b2m_conv_up, round 5): no LDS strip, three waves per SIMD by registers
    '_Z20conv_fwd_flow_kernelILi2ELi3ELi0ELi1ELi1ELi0ELi1EEv8ConvArgs': (168, 3, True),
    '_Z20conv_fwd_flow_kernelILi2ELi2ELi0ELi1ELi1ELi0ELi1EEv8ConvArgs': (168, 3, True),
    # ... and with hipcc-tracked loads (B2M_CONV_HANDLOADS=0, three steps in flight, diagnostics)
    '_Z20conv_fwd_flow_kernelILi2ELi3ELi0ELi1ELi0ELi0ELi0EEv8ConvArgs': (168, 3, True),
    '_Z20conv_fwd_flow_kernelILi2ELi2ELi0ELi1ELi0ELi0ELi0EEv8ConvArgs': (128, 4, True),
    # weight gradient with hand-issued loads (real rulebooks; every block shape of 2..4 x 2..4 sub-tiles): the shipped variants
    '_Z22conv_wgrad_flow_kernelILi2ELi2ELi1ELi0EEv9WgradArgs': (64, 8, True),
    '_Z22conv_wgrad_flow_kernelILi2ELi2ELi1ELi1EEv9WgradArgs': (64, 8, True),      # b2m_conv_wgrad_tr: the rulebook's row roles exchanged
    '_Z22conv_wgrad_flow_kernelILi2ELi3ELi1ELi0EEv9WgradArgs': (72, 7, True),
    '_Z22conv_wgrad_flow_kernelILi2ELi3ELi1ELi1EEv9WgradArgs': (72, 7, True),      # b2m_conv_wgrad_tr: the rulebook's row roles exchanged
    '_Z22conv_wgrad_flow_kernelILi3ELi2ELi1ELi0EEv9WgradArgs': (72, 7, True),
    '_Z22conv_wgrad_flow_kernelILi3ELi2ELi1ELi1EEv9WgradArgs': (72, 7, True),      # b2m_conv_wgrad_tr: the rulebook's row roles exchanged
    '_Z22conv_wgrad_flow_kernelILi2ELi4ELi1ELi0EEv9WgradArgs': (80, 6, True),
    '_Z22conv_wgrad_flow_kernelILi2ELi4ELi1ELi1EEv9WgradArgs': (80, 6, True),      # b2m_conv_wgrad_tr: the rulebook's row roles exchanged
    '_Z22conv_wgrad_flow_kernelILi4ELi2ELi1ELi0EEv9WgradArgs': (80, 6, True),
    '_Z22conv_wgrad_flow_kernelILi4ELi2ELi1ELi1EEv9WgradArgs': (80, 6, True),      # b2m_conv_wgrad_tr: the rulebook's row roles exchanged
    '_Z22conv_wgrad_flow_kernelILi3ELi3ELi1ELi0EEv9WgradArgs': (88, 5, True),
    '_Z22conv_wgrad_flow_kernelILi3ELi3ELi1ELi1EEv9WgradArgs': (88, 5, True),      # b2m_conv_wgrad_tr: the rulebook's row roles exchanged
    '_Z22conv_wgrad_flow_kernelILi3ELi4ELi1ELi0EEv9WgradArgs': (128, 4, True),
    '_Z22conv_wgrad_flow_kernelILi3ELi4ELi1ELi1EEv9WgradArgs': (128, 4, True),      # b2m_conv_wgrad_tr: the rulebook's row roles exchanged
    '_Z22conv_wgrad_flow_kernelILi4ELi3ELi1ELi0EEv9WgradArgs': (128, 4, True),
    '_Z22conv_wgrad_flow_kernelILi4ELi3ELi1ELi1EEv9WgradArgs': (128, 4, True),      # b2m_conv_wgrad_tr: the rulebook's row roles exchanged
    '_Z22conv_wgrad_flow_kernelILi4ELi4ELi1ELi0EEv9WgradArgs': (128, 4, True),
    '_Z22conv_wgrad_flow_kernelILi4ELi4ELi1ELi1EEv9WgradArgs': (128, 4, True),      # b2m_conv_wgrad_tr: the rulebook's row roles exchanged
    # ... hipcc-tracked loads (identity maps, B2M_WGRAD_HANDLOADS=0)
    '_Z22conv_wgrad_flow_kernelILi3ELi3ELi0ELi0EEv9WgradArgs': (88, 5, True),
    '_Z22conv_wgrad_flow_kernelILi4ELi4ELi0ELi0EEv9WgradArgs': (128, 4, True),
    '_Z22conv_wgrad_flow_kernelILi2ELi2ELi0ELi0EEv9WgradArgs': (64, 8, True),
    '_Z16conv_stem_kernelILb0EEv8ConvArgs': (128, 4, True),                       # the 5x5x5 first layer
    # half weight gradient on the f16 MFMA (round 6; hipcc-tracked loads, builtin MFMAs: accumulators in AGPRs, occupancy by LDS)
    '_Z21conv_wgrad_trh_kernelILi3ELi3ELi0EEv9WgradArgs': (64, 5, False),
    '_Z21conv_wgrad_trh_kernelILi3ELi3ELi1EEv9WgradArgs': (64, 5, False),
    '_Z21conv_wgrad_trh_kernelILi2ELi2ELi0EEv9WgradArgs': (48, 6, False),
    '_Z21conv_wgrad_trh_kernelILi4ELi4ELi0EEv9WgradArgs': (80, 3, False),
    '_Z21conv_wgrad_trh_kernelILi2ELi3ELi0EEv9WgradArgs': (56, 6, False),
    '_Z15conv_1x1_kernelILi3EEv8ConvArgs': (176, 2, False),                        # 1x1 streaming GEMM (compiler-scheduled waits)
    '_Z15conv_1x1_kernelILi2EEv8ConvArgs': (168, 3, False),
}


@pytest.mark.parametrize('name', sorted(PINNED))
def test_hot_kernel_resources(kernels, name):
    assert name in kernels, 'kernel not found in the device assembly (renamed / template arguments changed?): %s' % name
    k = kernels[name]
    max_vgpr, occ, counted = PINNED[name]
    assert k['vgpr'] <= max_vgpr, (name, k)
    assert k.get('scratch', 0) == 0, 'register spills to scratch in %s: %s' % (name, k)
    assert k['occupancy'] >= occ, (name, k)
    assert k['mfma'] > 0
    if counted:
        assert k['loop_waits'], name
        assert 0 not in k['loop_waits'], 's_waitcnt vmcnt(0) inside the MFMA loop of %s: %s' % (name, k['loop_waits'])


def test_hand_issued_loads_of_the_flow_kernel(kernels):
    """conv_fwd_flow_kernel<.., HL = 1> issues its operand loads as asm statements hipcc cannot see: (i) every load of an
    absent row group sits between an EXEC write and its restore, (ii) the waits in front of the MFMA blocks are the counted
    ones of conv_fwd_flow.h -- (D - 1) * (NG + TW) + g -- and the pair-list wait is D * (NG + TW), (iii) the loads still in
    flight when the offset loop ends are drained (vmcnt(0)) before the registers are reused for the write-out."""
    import isa_check
    text = open(isa_check.device_asm()).read()
    for tw, name in ((3, '_Z20conv_fwd_flow_kernelILi2ELi3ELi0ELi1ELi1ELi0ELi0EEv8ConvArgs'), (2, '_Z20conv_fwd_flow_kernelILi2ELi2ELi0ELi1ELi1ELi0ELi0EEv8ConvArgs')):
        body = text[text.index(name + ':'):]
        body = body[:body.index('.end_amdhsa_kernel')].split('\n')
        mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
        masked = [i for i, l in enumerate(body) if 's_mov_b64 exec, s[' in l]
        assert len(masked) >= 16                                  # 4 gathers x 2 buffers, prologue + loop
        for i in masked:
            assert 'global_load_dwordx4' in body[i + 1] and 's_mov_b64 exec, -1' in body[i + 2], body[i:i + 3]
        waits = kernels[name]['loop_waits']
        base = 4 + tw
        assert set(waits) <= {base, base + 1, base + 2, base + 3, 2 * base}, waits
        after = body[mf[-1]:]
        assert any('s_waitcnt vmcnt(0)' in l for l in after[:400]), 'no drain of the in-flight loads behind the offset loop'


HAND_ISSUED = [n for n in sorted(PINNED) if ('conv_fwd_flow' in n and 'ELi0ELi0ELi0EEv8' not in n) or re.search(r'Li1ELi[01]EEv9WgradArgs$', n)]


@pytest.mark.parametrize('name', HAND_ISSUED)
def test_nothing_touches_a_register_of_a_load_in_flight(kernels, name):
    """A hand-issued load fills its registers some hundred cycles after hipcc thinks they are written.  Between the load and
    the counted wait that covers it nothing may read or write them: tools/isa_check.py traces the assembly with the in-order
    load queue the counts refer to.  (Round 4: handed the loaded registers as in/out operands of a bare `s_waitcnt`, hipcc
    copied them to fresh registers in front of the wait -- whole 48 x 48 blocks of dW came out wrong on some runs.  The cure is
    in the source -- the statement that waits is the only reader -- and this is the check that it stays cured; the trace also
    proves the counts themselves: an MFMA reading an operand whose load is still among the N youngest is reported.)"""
    import isa_check
    assert len(HAND_ISSUED) == 36, HAND_ISSUED      # 18 forward variants (round 6: + the two-slice form, + four 64-column half variants) + 9 weight-gradient block shapes x {plain, exchanged row roles}
    body = isa_check.kernel_body(isa_check.device_asm(), name)
    assert sum(1 for i, l in enumerate(body) if 'global_load' in l and 'ASMSTART' in body[i - 1] + body[i - 2] + body[i - 3]) >= 8
    assert isa_check.inflight_violations(body) == []


def test_the_register_trace_sees_the_hazards_it_is_there_for():
    import isa_check
    asm = """
.LBB0_1: ; Loop Header
	;;#ASMSTART
	global_load_dword v10, v2, s[0:1]
	global_load_dword v11, v2, s[0:1] offset:64
	;;#ASMEND
	;;#ASMSTART
	global_load_dword v12, v3, s[0:1]
	;;#ASMEND
	%s
	;;#ASMSTART
	s_waitcnt vmcnt(%d)
	;;#ASMEND
	;;#ASMSTART
	v_mfma_f32_16x16x4_f32 v[20:23], v10, v11, v[20:23]
	;;#ASMEND
	s_cbranch_scc1 .LBB0_1
	s_endpgm
"""
    ok = isa_check.inflight_violations((asm % ('s_nop 0', 1)).split('\n'))
    assert ok == []
    copied = isa_check.inflight_violations((asm % ('v_mov_b32_e32 v30, v11', 1)).split('\n'))          # a copy in front of the wait
    assert [v[2] for v in copied] == [[11]]
    clobber = isa_check.inflight_violations((asm % ('v_add_u32_e32 v12, 4, v3', 1)).split('\n'))       # the allocator reuses v12
    assert [v[2] for v in clobber] == [[12]]
    short = isa_check.inflight_violations((asm % ('s_nop 0', 2)).split('\n'))                          # a wait that is one short
    assert [v[2] for v in short] == [[11]] and 'v_mfma' in short[0][1]


def test_the_register_trace_follows_the_path_that_skips_a_wait():
    """The consumer of a load (wait + MFMAs) sits behind a wave-uniform branch -- a k-step without pairs, an absent row group.
    On the path that jumps over it the load is still in flight when the next address is computed: a temporary in the load's
    register is overwritten when the data lands (round 4: a GPU memory fault in conv_wgrad_flow_kernel; latent in the
    conv_fwd_flow_kernel of the same round).  The cure pinned by test_nothing_touches_a_register_of_a_load_in_flight: load
    destinations are in/out operands, so the register is never free."""
    import isa_check
    asm = """
.LBB0_1: ; Loop Header
	;;#ASMSTART
	global_load_dword v10, v2, s[0:1]
	;;#ASMEND
	s_cbranch_scc1 .LBB0_3
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	;;#ASMSTART
	v_mfma_f32_16x16x4_f32 v[20:23], v10, v11, v[20:23]
	;;#ASMEND
.LBB0_3:
	%s
	s_cbranch_scc0 .LBB0_1
	s_endpgm
"""
    assert isa_check.inflight_violations((asm % 'v_add_u32_e32 v2, 4, v2').split('\n')) == []
    hit = isa_check.inflight_violations((asm % 'v_lshrrev_b32_e32 v10, 24, v3').split('\n'))     # the "dead" register as a temporary
    assert [v[2] for v in hit] == [[10]]


def test_no_mfma_reads_an_operand_a_valu_instruction_has_just_written(kernels):
    """The flow kernels' MFMAs are asm statements: hipcc pads no hazard for them.  A VGPR written by a VALU instruction (the
    select that zeroes a missing pair's dy, the half form's conversions) is read OLD by an MFMA issued fewer than two wait
    states later.  Round 6: conv_wgrad_flow_h_kernel<2, 2> -- `v_cndmask v29; v_cndmask v28; v_mfma .., v26, v29` -- added the
    UNMASKED dy of missing pairs into the first sub-tile of every block (dW 4 - 40 % too large at the offsets with few pairs),
    while the 3 x 2 block, one conversion further from its first MFMA, was exact; the fp32 form was one `s_waitcnt` away from
    the same.  The cure is `s_nop 1` in the first MFMA statement of a k-step; this is the check on EVERY kernel with MFMAs."""
    import isa_check
    path = isa_check.device_asm()
    checked = 0
    for name in sorted(kernels):
        if kernels[name]['mfma']:
            assert isa_check.mfma_operand_violations(isa_check.kernel_body(path, name)) == [], name
            checked += 1
    assert checked >= 150


def test_the_operand_check_sees_the_hazard_it_is_there_for():
    import isa_check
    bad = """
	v_cvt_f32_f16_e32 v26, v26
	v_cndmask_b32_e32 v29, 0, v29, vcc
	v_cndmask_b32_e32 v28, 0, v28, vcc
	;;#ASMSTART
	v_mfma_f32_16x16x4_f32 v[12:15], v26, v29, v[12:15]
	;;#ASMEND
	;;#ASMSTART
	v_mfma_f32_16x16x4_f32 v[8:11], v26, v28, v[8:11]
	;;#ASMEND
""".split('\n')
    v = isa_check.mfma_operand_violations(bad)
    assert [x[2] for x in v] == [[29], [28]]
    good = [l for l in bad]
    good.insert(good.index('\t;;#ASMSTART') + 1, '\ts_nop 1')
    assert isa_check.mfma_operand_violations(good) == []
    # a wide accumulator written by a VALU move and read as A two instructions later: still too close by one state
    assert isa_check.mfma_operand_violations(['v_mov_b32_e32 v4, 0', 's_waitcnt vmcnt(3)', 'v_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]']) != []
    assert isa_check.mfma_operand_violations(['v_mov_b32_e32 v4, 0', 's_nop 0', 's_waitcnt vmcnt(3)', 'v_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]']) == []


def test_half_weight_gradient_reads_its_operands_transposed(kernels):
    """conv_wgrad_trh_kernel: per 16-pair slot MI + NJ `ds_read_b64_tr_b16` (gfx950's transposing LDS read) feed MI x NJ
    v_mfma_f32_16x16x16_f16 whose accumulators stay in AGPRs for the whole walk -- no accumulator copies inside the loop (what the
    fp32 kernels' builtin form suffered from in round 1), 16-byte gathers, 16-byte LDS writes."""
    import isa_check
    path = isa_check.device_asm()
    for mi, nj in ((3, 3), (2, 2), (4, 4), (2, 4)):
        body = isa_check.kernel_body(path, '_Z21conv_wgrad_trh_kernelILi%dELi%dELi0EEv9WgradArgs' % (mi, nj))
        text = '\n'.join(body)
        assert text.count('ds_read_b64_tr_b16') == mi + nj, (mi, nj)
        mf = [l for l in body if 'v_mfma' in l]
        assert len(mf) == mi * nj and all('v_mfma_f32_16x16x16_f16 a[' in l for l in mf), mf[:2]
        first = next(i for i, l in enumerate(body) if 'v_mfma' in l)
        loop_top = max(i for i in range(first) if 'Loop Header' in body[i])
        back = max(i for i, l in enumerate(body) if re.search(r's_cbranch\w*\s+' + re.escape(body[loop_top].split(':')[0]), l) or
                   re.search(r's_branch\s+' + re.escape(body[loop_top].split(':')[0]), l))
        loop = body[loop_top:back + 1]
        assert not any('v_accvgpr' in l for l in loop), [l for l in loop if 'v_accvgpr' in l][:3]
        assert any('global_load_dwordx4' in l for l in loop) and any('ds_write_b128' in l for l in loop)
        assert not any('global_load_ushort' in l or 'global_load_short' in l for l in loop)
