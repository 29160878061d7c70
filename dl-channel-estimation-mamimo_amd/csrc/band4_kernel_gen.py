#!/usr/bin/env python3
"""band4_kernel_gen.py - gfx950 assembly of the REGISTER-BLOCKED bf16 band kernel ("csi_band4_bf16", round 6).

Same function, same arguments (Band8Args), same operand buffers as csi_band8_bf16 (band_kernel_gen.py):

    h1[m][k]  = bf16(relu(L0[m / nt][k] + Ts[m % nt][k]))
    h2[m][n]  = bf16(relu(sum_k h1[m][k] * W1[n][k] + bias1[n]))          (massiveMIMO_CSI_prediction_DNN.py:211-227)
    out[m][o] = sum_n h2[m][n] * W2[o][n] + bias2[o]

one workgroup per band of 128 pair rows, h2 in registers.  What differs is the blocking (round-5 verdict, next 2):

  4 waves x 512 registers, ONE wave per SIMD.  Wave w: feature half h = w >> 1, row pair rp = w & 1 (row groups 2 rp, 2 rp + 1 =
  rows 64 rp .. + 63 of the band).  A wave multiplies every weight fragment it reads against TWO row groups, so per MFMA the
  workgroup reads half the weight fragments from LDS (32 KiB per 32-k sub-step instead of 64) and issues its LDS-DMA from half
  as many waves.  Stage-1 accumulators (2 row groups x 4 feature tiles x 16) live in the ARCH half of the register file
  (v128 .. v255), so the h2 conversion reads them with v_pk_add_f32 directly - no v_accvgpr_read; the regressor's accumulators
  (128) and every operand that only ds_read writes and only MFMAs read (weight fragments, the partner's activation fragments,
  the h2 fragments) live in AGPRs (a0 .. a207).

  With one wave per SIMD nothing covers a wave's non-MFMA work but its own MFMAs: every sub-step is written as 8 + 8 MFMAs
  around its one barrier and the other instructions are DEALT OUT between them (<= 5 per gap is what an MFMA of 32 cycles
  hides, MI355X_MICROARCH.md).  To keep that stream even, the h2 fragments alternate between the halves (fragment q belongs to
  half q & 1: tiles 0, 4, 1, 5, ... of the column step - the regressor's k-order is free) and their conversion is cut into
  quarters (row group x k-step, 15 instructions) of which each half-sub-step carries one.

The vmcnt waits are counted by the same path simulation as band_kernel_gen.py (imported).

Usage: band4_kernel_gen.py out.s [variant ...]
"""
import re
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from band_kernel_gen import (Block, Wait, simulate, sreg, vreg, areg, DESCRIPTOR, S_AS2, S_INSC, S_AS1OS, S_OUTSC, META_KERNEL, KARG_BYTES, KARG_BYTES_CS, S_PART,     # noqa: E402
                             S_L0, S_TS, S_W1, S_B1, S_W2, S_B2, S_OUT, S_PEAK, S_STAMPS, S_LDL, S_NT, S_LDB1, S_M, S_K1, S_N1, S_LDB2, S_N2,
                             S_LDO, S_MAGIC, S_WAVE, S_H, S_M0, S_W1P, S_W2P, S_L0P, S_TSP, S_TRIP, S_COL, S_NCOL, S_NSUB1, S_DMA, S_T,
                             S_ROWMASK, S_SAVE, S_COLBYTES, S_BIAS2OFF, S_TSLABB)

# ----------------------------------------------------------------------------------------------- layout
RING_SLOT = 16384
AX1_OFF = 4 * RING_SLOT             # stage-1 fragment exchange  [rp][parity][r][k-step][64 lanes x 16 B] = 16 KiB
AX2_OFF = AX1_OFF + 16384           # h2 fragments               [rp][parity][r][k-step][64 lanes x 16 B] = 16 KiB
BIAS1_OFF = AX2_OFF + 16384
MAX_N1 = 4096
TS_OFF = BIAS1_OFF + 4 * MAX_N1 + 1024
TSLAB = 8192
L0S_OFF = TS_OFF + 4 * TSLAB
LDS_BYTES = L0S_OFF + 4 * 1024      # 152576 <= 163840

S_RP = 37                           # row pair of this wave (band_kernel_gen's S_RG)
S_TCH0, S_TCH1 = 73, 71             # byte offsets of this wave's two 1-KiB chunks of a T slab
S_ROWMASK1 = 76

V_TID, V_LANE = 0, 1
# Register plan.  Accumulators in AGPRs (a0 .. a127 the regressor's, a128 .. a255 stage 1's), MFMA A / B operands in LOW arch VGPRs, everything the
# VALU / LDS / DMA instructions touch in arch VGPRs from v128 up.  Two things about the MFMA operand fetch were measured on the skeleton
# (profiles/r06_band_probe.txt): (1) C / D in arch VGPRs with A / B in AGPRs issues ~15 % slower than any other combination (the first form of this
# kernel kept the stage-1 accumulators in arch VGPRs so that the h2 conversion could read them without v_accvgpr_read); (2) so do A / B operands in
# v128 .. v207 beside accumulators in a128 .. a255 - the operand registers want to sit below the accumulators' index range.
AW, OWNF, AF, F2 = 2, 34, 50, 66    # weight fragments [k-step][jj] x 4 | own stage-1 fragments [parity][r] x 4 | the partner's [parity][r] x 4 | h2 fragments [parity][r][k-step] x 4
SLT, SLL = 128, 136                 # slab values of the fragment being converted: T [r] x 8 at 128 + 16 r, L0 at 136 + 16 r
GV, BQ, CV = 160, 168, 184          # fp32 sums (8) / bias quads [2 sets] x 8 / converted h2 quarters [3] x 4
V_RD0, V_RD1, V_AX1, V_AX2 = 196, 197, 198, 199
V_TL0, V_TL1, V_LL = 200, 202, 204  # [r]
V_LOFF, V_TOFF = 206, 207           # V_TOFF [2]
V_VO = 209                          # LDS-DMA offset of this lane inside a weight sub-tile: 4096 w + 16 lane
V_BADDR, V_B2ADDR, V_OUTOFF, V_M, V_4HI, V_HI, V_L31 = 210, 211, 212, 214, 216, 217, 218
V_T = 128                           # scratch of prologue / epilogue / stamps: aliases the slab values (dead there)
ACC1 = 128                          # AGPRs a128 .. a255: stage-1 accumulators [r][jj] x 16
ACC2 = 0                            # AGPRs a0 .. a127: the regressor's [r][jj] x 16


def stamp(b, i, uid):
    """wave 0, lane 0: (shader cycles, wall ticks) into stamps[(workgroup * 6 + i) * 2 ..] (the library passes no stamp buffer)"""
    skip = 'L_stamp_skip_%d' % uid
    b.e('s_cmp_eq_u64 %s, 0' % sreg(S_STAMPS, 2))
    b.e('s_cbranch_scc1 %s' % skip)
    b.e('s_cmp_lg_u32 %s, 0' % sreg(S_WAVE))
    b.e('s_cbranch_scc1 %s' % skip)
    b.e('s_memtime %s' % sreg(S_T, 2))
    b.e('s_memrealtime %s' % sreg(S_T + 2, 2))
    b.e('s_mul_i32 %s, s2, 96' % sreg(S_T + 4))
    b.e('s_waitcnt lgkmcnt(0)')
    b.e('s_mov_b64 %s, exec' % sreg(S_SAVE, 2))
    b.e('s_mov_b64 exec, 1')
    for k in range(4):
        b.e('v_mov_b32_e32 %s, %s' % (vreg(V_T + k), sreg(S_T + k)))
    b.e('v_mov_b32_e32 %s, %s' % (vreg(V_T + 4), sreg(S_T + 4)))
    b.e('global_store_dwordx4 %s, %s, %s offset:%d' % (vreg(V_T + 4), vreg(V_T, 4), sreg(S_STAMPS, 2), 16 * i))
    b.e('s_mov_b64 exec, %s' % sreg(S_SAVE, 2))
    b.label(skip)


class Role4:
    """straight-line program of the waves of one feature half (h = 0 / 1)"""

    def __init__(self, h, dbg):
        self.h = h
        self.dbg = dbg
        self.uid = 0

    # ---------------------------------------------------------------- operands
    def w(self, s, jj):
        return vreg(AW + 16 * s + 4 * jj, 4)

    def f1(self, par, r, s):
        return vreg(OWNF + 8 * par + 4 * r, 4) if s == self.h else vreg(AF + 8 * par + 4 * r, 4)

    def f2(self, par, r, s):
        return vreg(F2 + 16 * par + 8 * r + 4 * s, 4)

    def mfma(self, kind, r, jj, s, par, zero_c=False):
        if kind == 1:
            d = areg(ACC1 + 16 * (4 * r + jj), 16)
            f = self.f1(par, r, s)
        else:
            d = areg(ACC2 + 16 * (4 * r + jj), 16)
            f = self.f2(par, r, s)
        return ['  v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s' % (d, self.w(s, jj), f, '0' if zero_c else d)]

    # ---------------------------------------------------------------- units of filler work (each: a list of items kept together)
    def u_read_w(self, s, slot):
        if 'noread' in self.dbg:
            return []
        return [['  ds_read_b128 %s, %s offset:%d' % (self.w(s, jj), vreg(V_RD1 if s else V_RD0), slot * RING_SLOT + jj * 2048)] for jj in range(4)]

    def u_pieces(self, kind, slot, pre=None):
        """this wave's 4 LDS-DMA pieces (1 KiB each) of the next sub-tile of stream `kind` into ring slot `slot`.  The weights come PRE-TILED
        (band4_tile_kernel: every 16-KiB sub-tile contiguous, in stream order, already in the ring's swizzled image): a piece is 1 KiB of
        whole 128-byte lines, and since the instruction's immediate offset moves the LDS destination together with the source
        (tools/ldsdma_offset_probe.hip) the four pieces share one m0 and one offset register"""
        ptr = S_W1P if kind == 's1' else S_W2P
        units = []
        if 'nodma' not in self.dbg:
            for p in range(4):
                u = ['  s_add_u32 m0, %s, %d' % (sreg(S_DMA), slot * RING_SLOT), '  s_nop 0'] if p == 0 else []
                u += ['  global_load_lds_dwordx4 %s, %s%s' % (vreg(V_VO), sreg(ptr, 2), ' offset:%d' % (1024 * p) if p else ''), ('vm', 'P%d' % slot)]
                units.append(u)
        units.append(['  s_add_u32 %s, %s, %d' % (sreg(ptr), sreg(ptr), RING_SLOT), '  s_addc_u32 %s, %s, 0' % (sreg(ptr + 1), sreg(ptr + 1))])
        return units

    def u_step(self, ptr, delta):
        if delta >= 0:
            return [['  s_add_u32 %s, %s, %d' % (sreg(ptr), sreg(ptr), delta), '  s_addc_u32 %s, %s, 0' % (sreg(ptr + 1), sreg(ptr + 1))]]
        return [['  s_sub_u32 %s, %s, %d' % (sreg(ptr), sreg(ptr), -delta), '  s_subb_u32 %s, %s, 0' % (sreg(ptr + 1), sreg(ptr + 1))]]

    def u_stage_issue(self, slot, reset=False):
        """LDS-DMA of the next T / L0 slab into slab slot `slot`: this wave's two 1-KiB chunks of the T slab, wave 0 the L0 slab"""
        units = []
        if reset:
            units.append(['  s_mov_b32 %s, %s' % (sreg(S_L0P), sreg(S_L0)), '  s_mov_b32 %s, %s' % (sreg(S_L0P + 1), sreg(S_L0 + 1)),
                          '  s_mov_b32 %s, %s' % (sreg(S_TSP), sreg(S_TS)), '  s_mov_b32 %s, %s' % (sreg(S_TSP + 1), sreg(S_TS + 1))])
        if 'noreq' not in self.dbg:
            for i, tch in enumerate((S_TCH0, S_TCH1)):
                units.append(['  s_add_u32 m0, %s, %d' % (sreg(tch), TS_OFF + slot * TSLAB), '  s_nop 0',
                              '  global_load_lds_dwordx4 %s, %s' % (vreg(V_TOFF + i), sreg(S_TSP, 2)), ('vm', 'T%d' % slot)])
            if self.h == 0:
                self.uid += 1
                skip = 'L_r0_l0skip_%d' % self.uid
                units.append(['  s_cmp_lg_u32 %s, 0' % sreg(S_WAVE), '  s_cbranch_scc1 %s' % skip, '  s_mov_b32 m0, %d' % (L0S_OFF + slot * 1024), '  s_nop 0',
                              '  global_load_lds_dwordx4 %s, %s' % (vreg(V_LOFF), sreg(S_L0P, 2)), ('vmopt', 'L%d' % slot), skip + ':'])
        units.append(['  s_add_u32 %s, %s, %s' % (sreg(S_TSP), sreg(S_TSP), sreg(S_TSLABB)), '  s_addc_u32 %s, %s, 0' % (sreg(S_TSP + 1), sreg(S_TSP + 1)),
                      '  s_add_u32 %s, %s, 128' % (sreg(S_L0P), sreg(S_L0P)), '  s_addc_u32 %s, %s, 0' % (sreg(S_L0P + 1), sreg(S_L0P + 1))])
        return units

    def u_stage_read(self, slot):
        """this lane's 8 T and 8 L0 values of its k-step (k-step = h) of the fragment whose slab sits in `slot`, both row groups"""
        if 'noreq' in self.dbg:
            return []
        units = []
        for r in range(2):
            units.append(['  ds_read_b128 %s, %s offset:%d' % (vreg(SLT + 16 * r, 4), vreg(V_TL0 + r), slot * TSLAB)])
            units.append(['  ds_read_b128 %s, %s offset:%d' % (vreg(SLT + 16 * r + 4, 4), vreg(V_TL1 + r), slot * TSLAB)])
            units.append(['  ds_read_b128 %s, %s offset:%d' % (vreg(SLL + 16 * r, 4), vreg(V_LL + r), slot * 1024)])
            units.append(['  ds_read_b128 %s, %s offset:%d' % (vreg(SLL + 16 * r + 4, 4), vreg(V_LL + r), slot * 1024 + 16)])
        return units

    def u_convert1(self, par_next):
        """own k-step (= h) of the next stage-1 fragment, both row groups: OWNF[par_next][r] and the exchange slot"""
        units = []
        for r in range(2):
            dst = OWNF + 8 * par_next + 4 * r
            if 'noconv' not in self.dbg:
                if 'pk' not in self.dbg:
                    for p in range(8):
                        units.append(['  v_add_f32_e32 %s, %s, %s' % (vreg(GV + p), vreg(SLL + 16 * r + p), vreg(SLT + 16 * r + p))])
                for p in range(4 if 'pk' in self.dbg else 0):
                    units.append(['  v_pk_add_f32 %s, %s, %s' % (vreg(GV + 2 * p, 2), vreg(SLL + 16 * r + 2 * p, 2), vreg(SLT + 16 * r + 2 * p, 2))])
                for p in range(4):
                    units.append(['  v_cvt_pk_bf16_f32 %s, %s, %s' % (vreg(dst + p), vreg(GV + 2 * p), vreg(GV + 2 * p + 1))])
                for p in range(4):
                    units.append(['  v_pk_max_i16 %s, %s, 0' % (vreg(dst + p), vreg(dst + p))])
            units.append(['  ds_write_b128 %s, %s offset:%d' % (vreg(V_AX1), vreg(dst, 4), par_next * 4096 + r * 2048 + self.h * 1024)])
        return units

    def u_read_f1(self, par_next):
        """the partner half's k-step of the next stage-1 fragment"""
        return [['  ds_read_b128 %s, %s offset:%d' % (vreg(AF + 8 * par_next + 4 * r, 4), vreg(V_AX1), par_next * 4096 + r * 2048 + (1 - self.h) * 1024)] for r in range(2)]

    def u_read_f2(self, par_next):
        return [['  ds_read_b128 %s, %s offset:%d' % (self.f2(par_next, r, s), vreg(V_AX2), par_next * 4096 + r * 2048 + s * 1024)] for r in range(2) for s in range(2)]

    # h2 quarter (fragment q, row group r, k-step s): accumulator registers 8 s .. 8 s + 7 of tile q >> 1 of this half
    def u_bias(self, q, s, bset):
        imm = (128 * self.h + 32 * (q >> 1) + 16 * s) * 4
        return [['  ds_read_b128 %s, %s offset:%d' % (vreg(BQ + 8 * bset, 4), vreg(V_BADDR), imm)],
                ['  ds_read_b128 %s, %s offset:%d' % (vreg(BQ + 8 * bset + 4, 4), vreg(V_BADDR), imm + 32)]]

    def u_quarter(self, q, r, s, bset, cset):
        if 'noconv' in self.dbg:
            return []
        acc = ACC1 + 16 * (4 * r + (q >> 1)) + 8 * s
        units = []
        for p in range(8):
            units.append(['  v_accvgpr_read_b32 %s, %s' % (vreg(GV + p), areg(acc + p))])
        if 'pk' not in self.dbg:
            for p in range(8):
                units.append(['  v_add_f32_e32 %s, %s, %s' % (vreg(GV + p), vreg(GV + p), vreg(BQ + 8 * bset + p))])
        for p in range(4 if 'pk' in self.dbg else 0):
            units.append(['  v_pk_add_f32 %s, %s, %s' % (vreg(GV + 2 * p, 2), vreg(GV + 2 * p, 2), vreg(BQ + 8 * bset + 2 * p, 2))])
        for p in range(4):
            units.append(['  v_cvt_pk_bf16_f32 %s, %s, %s' % (vreg(CV + 4 * cset + p), vreg(GV + 2 * p), vreg(GV + 2 * p + 1))])
        for p in range(4):
            units.append(['  v_pk_max_i16 %s, %s, 0' % (vreg(CV + 4 * cset + p), vreg(CV + 4 * cset + p))])
        return units

    def u_write_q(self, q, r, s, cset):
        return [['  ds_write_b128 %s, %s offset:%d' % (vreg(V_AX2), vreg(CV + 4 * cset, 4), (q & 1) * 4096 + r * 2048 + s * 1024)]]

    def barrier(self, b):
        if 'nobarrier' not in self.dbg:
            b.e('s_barrier')

    # ---------------------------------------------------------------- one sub-step: 8 MFMAs, barrier, 8 MFMAs, the rest dealt out between them
    @staticmethod
    def deal(b, mfmas, units):
        n, m = len(units), len(mfmas)
        done = 0
        for i, mf in enumerate(mfmas):
            b.items.extend(mf)
            upto = (n * (i + 1) + m - 1) // m if i < m - 1 else n
            for u in units[done:upto]:
                b.items.extend(u)
            done = max(done, upto)

    def substep(self, b, kind, slot, par, pre, post, needs, first=False):
        """kind 1 / 2: accumulator set.  P0 = k-step 1 (its weight fragments were read behind the previous barrier), P1 = k-step 0"""
        b.e('s_waitcnt lgkmcnt(0)')
        p0 = [self.mfma(kind, r, jj, 1, par, zero_c=first) for jj in range(4) for r in range(2)]
        p1 = [self.mfma(kind, r, jj, 0, par) for jj in range(4) for r in range(2)]
        if 'nointerleave' in self.dbg:
            for u in pre:
                b.items.extend(u)
            for mf in p0:
                b.items.extend(mf)
        else:
            self.deal(b, p0, pre)
        b.e('s_waitcnt lgkmcnt(0)')
        b.wait_vm(needs)
        self.barrier(b)
        if 'nointerleave' in self.dbg:
            for u in post:
                b.items.extend(u)
            for mf in p1:
                b.items.extend(mf)
        else:
            self.deal(b, p1, post)

    # ---------------------------------------------------------------- the role's program
    def build(self):
        h = self.h
        L = lambda s: 'L_r%d_%s' % (h, s)
        pro, head, loop, last, bub, st2, tail = (Block(n) for n in ('pro', 'head', 'loop', 'last', 'bub', 'st2', 'tail'))

        def flat(b, units):
            for u in units:
                b.items.extend(u)

        # ---- prologue
        b = pro
        b.label(L('start'))
        b.e('s_mov_b32 %s, %s' % (sreg(S_W2P), sreg(S_W2)))
        b.e('s_mov_b32 %s, %s' % (sreg(S_W2P + 1), sreg(S_W2 + 1)))
        if 'nodma' not in self.dbg:                          # (the common prologue issued this wave's pieces of sub-tiles 0 .. 3 behind the table loads)
            for t in range(4):
                b.items.extend([('vm', 'P%d' % t)] * 4)
        if 'roleslabs' in self.dbg:                          # (A/B: the first four slabs requested here instead of in the common prologue)
            for t in range(4):
                flat(b, self.u_stage_issue(t, reset=(t == 0)))
        elif 'noreq' not in self.dbg:                        # (... and the first four T / L0 slabs)
            for t in range(4):
                b.items.extend([('vm', 'T%d' % t), ('vm', 'T%d' % t)] + ([('vmopt', 'L%d' % t)] if h == 0 else []))
        b.wait_vm({'T0', 'L0'})
        b.e('s_waitcnt lgkmcnt(0)')
        self.barrier(b)
        flat(b, self.u_stage_read(0))
        b.e('s_waitcnt lgkmcnt(0)')
        flat(b, self.u_convert1(0))
        b.wait_vm({'P0', 'T1', 'L1'})
        b.e('s_waitcnt lgkmcnt(0)')
        self.barrier(b)
        flat(b, self.u_read_f1(0))
        flat(b, self.u_read_w(1, 0))
        flat(b, self.u_stage_read(1))
        b.e('s_mov_b32 %s, 0' % sreg(S_COL))

        # ---- stage 1, sub-step u (i = u & 3): before the barrier the plane-0 weight reads, the own k-step of fragment u + 1 and the request
        # of slab u + 4 (slab u was read behind barrier u - 2); behind it the plane-1 reads of sub-tile u + 1, the partner's k-step of
        # fragment u + 1, the values of fragment u + 2 and this wave's pieces of sub-tile u + 4
        def s1_substep(b, i, first=False, produce_ok=True, piece='s1', stage_ok=True, issue_ok=True):
            slot, par, nslot = i & 3, i & 1, (i + 1) & 3
            pre = self.u_read_w(0, slot)
            if produce_ok:
                pre += self.u_convert1(par ^ 1)
            if issue_ok:
                pre += self.u_stage_issue(i & 3)
            post = self.u_read_w(1, nslot)
            if produce_ok:
                post += self.u_read_f1(par ^ 1)
            if stage_ok:
                post += self.u_stage_read((i + 2) & 3)
            # (piece 's2': the regressor's sub-tiles 0 .. 3 of this column step, in fragment order = tiles 0, 4, 1, 5)
            post += self.u_pieces(piece, slot)
            needs = {'P%d' % nslot}
            if stage_ok:
                needs |= {'T%d' % ((i + 2) & 3), 'L%d' % ((i + 2) & 3)}
            self.substep(b, 1, slot, par, pre, post, needs, first=first)

        head.label(L('col'))
        for i in range(4):
            s1_substep(head, i, first=(i == 0))
        head.e('s_lshr_b32 %s, %s, 2' % (sreg(S_TRIP), sreg(S_NSUB1)))
        head.e('s_sub_u32 %s, %s, 2' % (sreg(S_TRIP), sreg(S_TRIP)))
        head.e('s_cmp_eq_u32 %s, 0' % sreg(S_TRIP))
        head.e('s_cbranch_scc1 %s' % L('last'))
        loop.items.append('  .p2align 6')
        loop.label(L('loop'))
        for i in range(4):
            s1_substep(loop, i)
        loop.e('s_sub_u32 %s, %s, 1' % (sreg(S_TRIP), sreg(S_TRIP)))
        loop.e('s_cmp_lg_u32 %s, 0' % sreg(S_TRIP))
        loop.e('s_cbranch_scc1 %s' % L('loop'))
        last.label(L('last'))
        for i in range(4):
            s1_substep(last, i, produce_ok=(i < 3), piece='s2', stage_ok=(i < 2), issue_ok=False)

        # ---- between the stages: the accumulators are final.  Half 0 converts fragment 0 (its tile 0) and the first quarter of fragment 2,
        # half 1 fragment 1 (its tile 0); bias quads for the quarter of section 0
        b = bub
        b.e('s_nop 15')
        b.e('s_nop 15')
        q0 = h                                              # fragment 0 / 1
        if 'noconv' not in self.dbg:
            for s in range(2):
                flat(b, self.u_bias(q0, s, s))
        b.e('s_waitcnt lgkmcnt(0)')
        for r in range(2):
            for s in range(2):
                flat(b, self.u_quarter(q0, r, s, s, 0))
                flat(b, self.u_write_q(q0, r, s, 0))
        if h == 0 and 'noconv' not in self.dbg:             # fragment 2, quarter 0 (r 0, k-step 0): held in CV set 0 until barrier 0 has passed
            flat(b, self.u_bias(2, 0, 0))
            b.e('s_waitcnt lgkmcnt(0)')
            flat(b, self.u_quarter(2, 0, 0, 0, 0))
            flat(b, self.u_bias(2, 1, 0))                   # quarter 1 (section 0 = pre 0) wants bias set 0
        b.e('s_waitcnt lgkmcnt(0)')
        self.barrier(b)
        flat(b, self.u_read_f2(0))

        # ---- stage 2.  Sections: pre q = 2 q, post q = 2 q + 1.  Fragment f (owner f & 1, f >= 2): quarter 0 (r0 s0) in section 2 f - 5,
        # quarter 1 (r0 s1) in 2 f - 4, quarter 2 (r1 s0) in 2 f - 3 together with the writes of quarters 0 .. 2 (behind barrier f - 2: the
        # buffer's previous fragment f - 2 has been read by every wave), quarter 3 (r1 s1) + its write in 2 f - 2 (before barrier f - 1).
        # The bias quads of a section's quarter are read one section earlier (set = section & 1).
        NQ = 8
        quarters = {}                                       # section -> (fragment, quarter)
        for f in range(2 + h, NQ, 2):
            for k in range(4):
                quarters[2 * f - 5 + k] = (f, k)

        def conv_units(sec):
            units = []
            if sec in quarters:
                f, k = quarters[sec]
                r, s = k >> 1, k & 1
                units += self.u_quarter(f, r, s, sec & 1, k if k < 3 else 0)
                if k == 2:
                    for kk in range(3):
                        units += self.u_write_q(f, kk >> 1, kk & 1, kk)
                if k == 3:
                    units += self.u_write_q(f, 1, 1, 0)
            if sec + 1 in quarters and 'noconv' not in self.dbg:
                f, k = quarters[sec + 1]
                units += self.u_bias(f, k & 1, (sec + 1) & 1)
            return units

        for q in range(NQ):
            slot, par, nslot = q & 3, q & 1, (q + 1) & 3
            pre = self.u_read_w(0, slot) + conv_units(2 * q)
            post = self.u_read_w(1, nslot)
            needs = {'P%d' % nslot}
            if q < NQ - 1:
                post += self.u_read_f2(par ^ 1)
            # the next column step's T / L0 streams: slabs 0 .. 3 requested before barriers 3 .. 6, fragment 0 from slab 0 (read behind
            # barrier 6, converted before barrier 7), fragment 1's values read behind barrier 7
            if 3 <= q <= 6:
                pre += self.u_stage_issue(q - 3, reset=(q == 3))
            if q == NQ - 2:
                post += self.u_stage_read(0)
                needs |= {'T0', 'L0'}
            if q == NQ - 1:
                pre += self.u_convert1(0)
                post += self.u_read_f1(0) + self.u_stage_read(1)
                needs |= {'T1', 'L1'}
            post += conv_units(2 * q + 1)
            # sub-tiles 4 .. 7 of the regressor stream, then sub-tiles 0 .. 3 of the next column step's pair-layer stream (the tiled copy is
            # in stream order: the pointer just runs on; behind the last column step it runs into the copy's four spare tiles)
            post += self.u_pieces('s2' if q < NQ - 4 else 's1', slot)
            self.substep(st2, 2, slot, par, pre, post, needs)
        tail.e('v_add_u32_e32 %s, 1024, %s' % (vreg(V_BADDR), vreg(V_BADDR)))
        tail.e('s_add_u32 %s, %s, 1' % (sreg(S_COL), sreg(S_COL)))
        tail.e('s_cmp_lt_u32 %s, %s' % (sreg(S_COL), sreg(S_NCOL)))
        tail.e('s_cbranch_scc1 %s' % L('col'))
        tail.e('s_branch L_epilogue_%d' % h)

        col0 = [head, last, bub, st2, tail]
        col1 = [head, loop, last, bub, st2, tail]
        col2 = [head, loop, loop, last, bub, st2, tail]
        for first_col in (col0, col1, col2):
            for second_col in (col0, col1, col2):
                for with_opt in (True, False):
                    simulate([pro] + first_col + second_col + second_col, with_opt=with_opt)
        return [pro, head, loop, last, bub, st2, tail]


# =============================================================================================== split-f16 form ("csi_band4": fp32 contexts)
# Same blocking for the split engine of fp32 contexts (gemm_hs.hip.h operands: every value as hi + lo f16 halves, three MFMAs per product:
# w_lo x a_hi, w_hi x a_hi, w_hi x a_lo).  Sub-step = 16 k: the weight sub-tile [256 rows][16 k as hi | lo] has the bf16 form's geometry
# (plane 0 = hi, plane 1 = lo); an activation fragment is (hi, lo) of 32 rows x 16 k.  Wave (h, rp) CONVERTS the fragments of row group
# r = h of its row pair (both halves, the whole 16 k) and reads the partner's from the exchange; 24 MFMAs per sub-step and wave.
# Stage 2: 16 fragments per column step (tile x half g of its 16-feature groups), owner q & 1, order (tile jt = q >> 2, g = (q >> 1) & 1).
H_OWNF = OWNF                        # own stage-1 fragment: [parity][hi | lo] x 4
H_SLT, H_SLL, H_GV, H_BQ, H_CV = 128, 136, 144, 152, 160     # slab values T / L0 (8 each), fp32 temporaries (8), bias quads (8), converted h2 units [2][hi | lo] x 4
V_PK1, V_PK2 = 176, 177              # packed maxima of the hi halves (range guard)
H_AF = AF                            # the partner's stage-1 fragment [parity][hi | lo] x 4
H_F2 = F2                            # h2 fragments [parity][r][hi | lo] x 4


class RoleH(Role4):
    """split-f16 form: straight-line program of the waves of one feature half"""

    def wh(self, plane, jj):
        return vreg(AW + 16 * plane + 4 * jj, 4)

    def a1(self, par, r, plane):
        return vreg(H_OWNF + 8 * par + 4 * plane, 4) if r == self.h else vreg(H_AF + 8 * par + 4 * plane, 4)

    def a2(self, par, r, plane):
        return vreg(H_F2 + 16 * par + 8 * r + 4 * plane, 4)

    def mf(self, kind, r, jj, wplane, aplane, par, zero_c=False):
        if kind == 1:
            d, f = areg(ACC1 + 16 * (4 * r + jj), 16), self.a1(par, r, aplane)
        else:
            d, f = areg(ACC2 + 16 * (4 * r + jj), 16), self.a2(par, r, aplane)
        return ['  v_mfma_f32_32x32x16_f16 %s, %s, %s, %s' % (d, self.wh(wplane, jj), f, '0' if zero_c else d)]

    def u_read_w(self, plane, slot):
        if 'noread' in self.dbg:
            return []
        return [['  ds_read_b128 %s, %s offset:%d' % (self.wh(plane, jj), vreg(V_RD1 if plane else V_RD0), slot * RING_SLOT + jj * 2048)] for jj in range(4)]

    def u_stage_issue(self, slot, reset=False):
        units = []
        if reset:
            units.append(['  s_mov_b32 %s, %s' % (sreg(S_L0P), sreg(S_L0)), '  s_mov_b32 %s, %s' % (sreg(S_L0P + 1), sreg(S_L0 + 1)),
                          '  s_mov_b32 %s, %s' % (sreg(S_TSP), sreg(S_TS)), '  s_mov_b32 %s, %s' % (sreg(S_TSP + 1), sreg(S_TS + 1))])
        if 'noreq' not in self.dbg:
            for i, tch in enumerate((S_TCH0, S_TCH1)):
                units.append(['  s_add_u32 m0, %s, %d' % (sreg(tch), TS_OFF + slot * TSLAB), '  s_nop 0',
                              '  global_load_lds_dwordx4 %s, %s' % (vreg(V_TOFF + i), sreg(S_TSP, 2)), ('vm', 'T%d' % slot)])
            if self.h == 0:
                self.uid += 1
                skip = 'L_r0_l0skip_%d' % self.uid
                units.append(['  s_cmp_lg_u32 %s, 0' % sreg(S_WAVE), '  s_cbranch_scc1 %s' % skip, '  s_mov_b32 m0, %d' % (L0S_OFF + slot * 1024), '  s_nop 0',
                              '  global_load_lds_dwordx4 %s, %s' % (vreg(V_LOFF), sreg(S_L0P, 2)), ('vmopt', 'L%d' % slot), skip + ':'])
        units.append(['  s_add_u32 %s, %s, %s' % (sreg(S_TSP), sreg(S_TSP), sreg(S_TSLABB)), '  s_addc_u32 %s, %s, 0' % (sreg(S_TSP + 1), sreg(S_TSP + 1)),
                      '  s_add_u32 %s, %s, 64' % (sreg(S_L0P), sreg(S_L0P)), '  s_addc_u32 %s, %s, 0' % (sreg(S_L0P + 1), sreg(S_L0P + 1))])
        return units

    def u_stage_read(self, slot):
        """this lane's 8 T and 8 L0 values (k = 8 hi .. + 7 of the sub-step's 16) of ITS row of row group r = h"""
        if 'noreq' in self.dbg:
            return []
        return [['  ds_read_b128 %s, %s offset:%d' % (vreg(H_SLT, 4), vreg(V_TL0), slot * TSLAB)],
                ['  ds_read_b128 %s, %s offset:%d' % (vreg(H_SLT + 4, 4), vreg(V_TL1), slot * TSLAB)],
                ['  ds_read_b128 %s, %s offset:%d' % (vreg(H_SLL, 4), vreg(V_LL), slot * 1024)],
                ['  ds_read_b128 %s, %s offset:%d' % (vreg(H_SLL + 4, 4), vreg(V_LL), slot * 1024 + 16)]]

    def split_units(self, hi, lo, pk):
        """GV (8 fp32, relu applied) -> hi halves, lo = f16(x - hi) through v_fma_mix (gemm_hs.hip.h hs_lo_pair), packed maximum of the hi halves"""
        u = []
        for p in range(4):
            u.append(['  v_cvt_pk_f16_f32 %s, %s, %s' % (vreg(hi + p), vreg(H_GV + 2 * p), vreg(H_GV + 2 * p + 1))])
        for p in range(4):
            u.append(['  v_fma_mixlo_f16 %s, %s, -1.0, %s op_sel_hi:[1,0,0]' % (vreg(lo + p), vreg(hi + p), vreg(H_GV + 2 * p))])
            u.append(['  v_pk_max_u16 %s, %s, %s' % (vreg(pk), vreg(pk), vreg(hi + p))])
        for p in range(4):
            u.append(['  v_fma_mixhi_f16 %s, %s, -1.0, %s op_sel:[1,0,0] op_sel_hi:[1,0,0]' % (vreg(lo + p), vreg(hi + p), vreg(H_GV + 2 * p + 1))])
        return u

    def u_convert1(self, par_next):
        hi, lo = H_OWNF + 8 * par_next, H_OWNF + 8 * par_next + 4
        units = []
        if 'noconv' not in self.dbg:
            for e in range(8):
                units.append(['  v_fma_f32 %s, %s, %s, %s' % (vreg(H_GV + e), vreg(H_SLL + e), sreg(S_INSC), vreg(H_SLT + e))])
            for e in range(8):
                units.append(['  v_max_f32_e32 %s, 0, %s' % (vreg(H_GV + e), vreg(H_GV + e))])
            units += self.split_units(hi, lo, V_PK1)
        units.append(['  ds_write_b128 %s, %s offset:%d' % (vreg(V_AX1), vreg(hi, 4), par_next * 4096 + self.h * 2048)])
        units.append(['  ds_write_b128 %s, %s offset:%d' % (vreg(V_AX1), vreg(lo, 4), par_next * 4096 + self.h * 2048 + 1024)])
        return units

    def u_read_f1(self, par_next):
        o = 1 - self.h
        return [['  ds_read_b128 %s, %s offset:%d' % (vreg(H_AF + 8 * par_next + 4 * pl, 4), vreg(V_AX1), par_next * 4096 + o * 2048 + pl * 1024)] for pl in range(2)]

    def u_read_f2(self, par_next):
        return [['  ds_read_b128 %s, %s offset:%d' % (self.a2(par_next, r, pl), vreg(V_AX2), par_next * 4096 + r * 2048 + pl * 1024)] for r in range(2) for pl in range(2)]

    @staticmethod
    def frag(q):
        """fragment q of a column step -> (tile of its owner, half g of the tile's 16-feature groups)"""
        return q >> 2, (q >> 1) & 1

    def u_bias(self, q):
        jt, g = self.frag(q)
        imm = (128 * self.h + 32 * jt + 16 * g) * 4
        return [['  ds_read_b128 %s, %s offset:%d' % (vreg(H_BQ, 4), vreg(V_BADDR), imm)],
                ['  ds_read_b128 %s, %s offset:%d' % (vreg(H_BQ + 4, 4), vreg(V_BADDR), imm + 32)]]

    def u_unit(self, q, r, cset):
        """h2 unit (fragment q, row group r): accumulator registers 8 g .. of tile jt -> (hi, lo) in CV set cset"""
        if 'noconv' in self.dbg:
            return []
        jt, g = self.frag(q)
        acc = ACC1 + 16 * (4 * r + jt) + 8 * g
        u = []
        for e in range(8):
            u.append(['  v_accvgpr_read_b32 %s, %s' % (vreg(H_GV + e), areg(acc + e))])
        for e in range(8):
            u.append(['  v_fma_f32 %s, %s, %s, %s' % (vreg(H_GV + e), vreg(H_GV + e), sreg(S_AS1OS), vreg(H_BQ + e))])
        for e in range(8):
            u.append(['  v_max_f32_e32 %s, 0, %s' % (vreg(H_GV + e), vreg(H_GV + e))])
        return u + self.split_units(H_CV + 8 * cset, H_CV + 8 * cset + 4, V_PK2)

    def u_write_unit(self, q, r, cset):
        return [['  ds_write_b128 %s, %s offset:%d' % (vreg(V_AX2), vreg(H_CV + 8 * cset + 4 * pl, 4), (q & 1) * 4096 + r * 2048 + pl * 1024)] for pl in range(2)]

    def substep(self, b, kind, slot, par, pre, post, needs, first=False):
        """P0 = w_lo x a_hi (w_lo was read behind the previous barrier), P1 = w_hi x a_hi, barrier, P2 = w_hi x a_lo"""
        b.e('s_waitcnt lgkmcnt(0)')
        order = [(jj, r) for jj in range(4) for r in range(2)]
        p0 = [self.mf(kind, r, jj, 1, 0, par, zero_c=first) for jj, r in order]
        p1 = [self.mf(kind, r, jj, 0, 0, par) for jj, r in order]
        p2 = [self.mf(kind, r, jj, 0, 1, par) for jj, r in order]
        # the plane-0 (hi) weight reads lead the list: they go out among P0's first MFMAs and are waited for in front of P1
        cut = min(len(pre), max(4, (len(pre) + 1) // 2))
        self.deal(b, p0, pre[:cut])
        b.e('s_waitcnt lgkmcnt(0)')
        self.deal(b, p1, pre[cut:])
        b.e('s_waitcnt lgkmcnt(0)')
        b.wait_vm(needs)
        self.barrier(b)
        self.deal(b, p2, post)

    def build(self):
        h = self.h
        L = lambda s: 'L_r%d_%s' % (h, s)
        pro, head, loop, last, bub, st2, tail = (Block(n) for n in ('pro', 'head', 'loop', 'last', 'bub', 'st2', 'tail'))

        def flat(b, units):
            for u in units:
                b.items.extend(u)

        b = pro
        b.label(L('start'))
        b.e('s_mov_b32 %s, %s' % (sreg(S_W2P), sreg(S_W2)))
        b.e('s_mov_b32 %s, %s' % (sreg(S_W2P + 1), sreg(S_W2 + 1)))
        if 'nodma' not in self.dbg:                          # (the common prologue issued this wave's pieces of sub-tiles 0 .. 3 behind the table loads)
            for t in range(4):
                b.items.extend([('vm', 'P%d' % t)] * 4)
        if 'roleslabs' in self.dbg:                          # (A/B: the first four slabs requested here instead of in the common prologue)
            for t in range(4):
                flat(b, self.u_stage_issue(t, reset=(t == 0)))
        elif 'noreq' not in self.dbg:                        # (... and the first four T / L0 slabs)
            for t in range(4):
                b.items.extend([('vm', 'T%d' % t), ('vm', 'T%d' % t)] + ([('vmopt', 'L%d' % t)] if h == 0 else []))
        b.wait_vm({'T0', 'L0'})
        b.e('s_waitcnt lgkmcnt(0)')
        self.barrier(b)
        flat(b, self.u_stage_read(0))
        b.e('s_waitcnt lgkmcnt(0)')
        flat(b, self.u_convert1(0))
        b.wait_vm({'P0', 'T1', 'L1'})
        b.e('s_waitcnt lgkmcnt(0)')
        self.barrier(b)
        flat(b, self.u_read_f1(0))
        flat(b, self.u_read_w(1, 0))
        flat(b, self.u_stage_read(1))
        b.e('s_mov_b32 %s, 0' % sreg(S_COL))

        def s1_substep(b, i, first=False, produce_ok=True, piece='s1', stage_ok=True, issue_ok=True):
            slot, par, nslot = i & 3, i & 1, (i + 1) & 3
            pre = self.u_read_w(0, slot)
            if produce_ok:
                pre += self.u_convert1(par ^ 1)
            if issue_ok:
                pre += self.u_stage_issue(i & 3)
            post = self.u_read_w(1, nslot)
            if produce_ok:
                post += self.u_read_f1(par ^ 1)
            if stage_ok:
                post += self.u_stage_read((i + 2) & 3)
            post += self.u_pieces(piece, slot)
            needs = {'P%d' % nslot}
            if stage_ok:
                needs |= {'T%d' % ((i + 2) & 3), 'L%d' % ((i + 2) & 3)}
            self.substep(b, 1, slot, par, pre, post, needs, first=first)

        head.label(L('col'))
        for i in range(4):
            s1_substep(head, i, first=(i == 0))
        head.e('s_lshr_b32 %s, %s, 2' % (sreg(S_TRIP), sreg(S_NSUB1)))
        head.e('s_sub_u32 %s, %s, 2' % (sreg(S_TRIP), sreg(S_TRIP)))
        head.e('s_cmp_eq_u32 %s, 0' % sreg(S_TRIP))
        head.e('s_cbranch_scc1 %s' % L('last'))
        loop.items.append('  .p2align 6')
        loop.label(L('loop'))
        for i in range(4):
            s1_substep(loop, i)
        loop.e('s_sub_u32 %s, %s, 1' % (sreg(S_TRIP), sreg(S_TRIP)))
        loop.e('s_cmp_lg_u32 %s, 0' % sreg(S_TRIP))
        loop.e('s_cbranch_scc1 %s' % L('loop'))
        last.label(L('last'))
        for i in range(4):
            s1_substep(last, i, produce_ok=(i < 3), piece='s2', stage_ok=(i < 2), issue_ok=False)

        # ---- between the stages: half h converts fragment h (its tile 0, g 0) for both row groups
        b = bub
        b.e('s_nop 15')
        b.e('s_nop 15')
        if 'noconv' not in self.dbg:
            flat(b, self.u_bias(h))
        b.e('s_waitcnt lgkmcnt(0)')
        for r in range(2):
            flat(b, self.u_unit(h, r, 0))
            flat(b, self.u_write_unit(h, r, 0))
        if h == 0 and 'noconv' not in self.dbg:
            flat(b, self.u_bias(2))                          # fragment 2's units: pre 0 (r 0), pre 1 (r 1)
        b.e('s_waitcnt lgkmcnt(0)')
        self.barrier(b)
        flat(b, self.u_read_f2(0))

        # ---- stage 2: 16 sub-steps.  Fragment f >= 2 (owner f & 1): unit r 0 converted before barrier f - 2 and written behind it (the buffer's
        # previous fragment f - 2 has been read by every wave then), unit r 1 converted and written before barrier f - 1; the bias quads of a
        # fragment are read behind barrier f - 3 (one set: both units use them)
        NQ = 16
        for q in range(NQ):
            slot, par, nslot = q & 3, q & 1, (q + 1) & 3
            pre = self.u_read_w(0, slot)
            post = self.u_read_w(1, nslot)
            needs = {'P%d' % nslot}
            if q < NQ - 1:
                post += self.u_read_f2(par ^ 1)
            fa, fb = q + 2, q + 1                            # fragments whose unit r 0 / r 1 this half may own in this sub-step
            if fa < NQ and (fa & 1) == h and fa >= 2:
                pre += self.u_unit(fa, 0, 0)
                post += self.u_write_unit(fa, 0, 0)
            if fb < NQ and (fb & 1) == h and fb >= 2:
                pre += self.u_unit(fb, 1, 1) + self.u_write_unit(fb, 1, 1)
            fn = q + 3                                       # bias quads of the fragment whose first unit comes in the next sub-step
            if fn < NQ and (fn & 1) == h and 'noconv' not in self.dbg:
                post += self.u_bias(fn)
            if NQ - 5 <= q <= NQ - 2:
                pre += self.u_stage_issue(q - (NQ - 5), reset=(q == NQ - 5))
            if q == NQ - 2:
                post += self.u_stage_read(0)
                needs |= {'T0', 'L0'}
            if q == NQ - 1:
                pre += self.u_convert1(0)
                post += self.u_read_f1(0) + self.u_stage_read(1)
                needs |= {'T1', 'L1'}
            post += self.u_pieces('s2' if q < NQ - 4 else 's1', slot)
            self.substep(st2, 2, slot, par, pre, post, needs)
        tail.e('v_add_u32_e32 %s, 1024, %s' % (vreg(V_BADDR), vreg(V_BADDR)))
        tail.e('s_add_u32 %s, %s, 1' % (sreg(S_COL), sreg(S_COL)))
        tail.e('s_cmp_lt_u32 %s, %s' % (sreg(S_COL), sreg(S_NCOL)))
        tail.e('s_cbranch_scc1 %s' % L('col'))
        tail.e('s_branch L_epilogue_%d' % h)

        col0 = [head, last, bub, st2, tail]
        col1 = [head, loop, last, bub, st2, tail]
        col2 = [head, loop, loop, last, bub, st2, tail]
        for first_col in (col0, col1, col2):
            for second_col in (col0, col1, col2):
                for with_opt in (True, False):
                    simulate([pro] + first_col + second_col + second_col, with_opt=with_opt)
        return [pro, head, loop, last, bub, st2, tail]


def common_prologue(b, dbg=(), hs=False):
    if 'persist' in dbg:
        # persistent form ("csi_band4_p" / "csi_band4_bf16_p"): a launch of P workgroups, workgroup x computes bands x, x + P, x + 2 P, ...: behind a band's
        # stores it comes back HERE with s2 += P (the argument record is read again: the body moves some of its pointers).  The record is the column-split
        # one (144 bytes); its last quadword holds P.  Outstanding stores at the restart are older than every new operation, so the counted vmcnt waits
        # of the body only become conservative by them.
        b.label('L_restart')
        b.e('s_load_dwordx2 %s, s[0:1], 0x88' % sreg(S_PART, 2))
    b.e('s_load_dwordx16 %s, s[0:1], 0x0' % sreg(4, 16))
    b.e('s_load_dwordx16 %s, s[0:1], 0x40' % sreg(20, 16))
    b.e('s_waitcnt lgkmcnt(0)')
    b.e('v_and_b32_e32 %s, 63, v0' % vreg(V_LANE))
    b.e('v_readfirstlane_b32 %s, v0' % sreg(S_WAVE))
    b.e('s_nop 4')
    b.e('s_lshr_b32 %s, %s, 6' % (sreg(S_WAVE), sreg(S_WAVE)))
    b.e('s_and_b32 %s, %s, 1' % (sreg(S_RP), sreg(S_WAVE)))
    b.e('s_lshr_b32 %s, %s, 1' % (sreg(S_H), sreg(S_WAVE)))
    b.e('s_lshl_b32 %s, s2, 7' % sreg(S_M0))
    b.e('s_cmp_ge_i32 %s, %s' % (sreg(S_M0), sreg(S_M)))
    b.e('s_cbranch_scc1 L_end')
    if 'colsplit' in dbg:
        # column-split form (small calls: fewer bands than CUs, as csi_band8_cs): workgroup (x, y) computes band x over the N1 hidden features
        # [y N1, (y + 1) N1) of a layer gridDim.y * N1 wide.  The arguments describe split 0; split y moves the (pre-tiled, stream-ordered) pair-layer
        # weights by its N1 / 256 column steps of K1 / SK sub-tiles, the regressor's by N1 / 256 x NQ sub-tiles, bias1 by y N1 entries, reads a zero bias2
        # and writes its partial outputs to part + (y - 1) * M * ldo * 4; the host adds the partials to split 0's output in y order
        b.e('s_load_dwordx2 %s, s[0:1], 0x80' % sreg(S_PART, 2))
        b.e('s_waitcnt lgkmcnt(0)')
        b.e('s_cmp_eq_u32 s3, 0')
        b.e('s_cbranch_scc1 L_cs_done')
        b.e('s_lshr_b32 %s, %s, 8' % (sreg(S_T), sreg(S_N1)))                            # column steps of one split
        b.e('s_mul_i32 %s, %s, s3' % (sreg(S_T), sreg(S_T)))                              # ... in front of this split
        b.e('s_lshr_b32 %s, %s, %d' % (sreg(S_T + 1), sreg(S_K1), 4 if hs else 5))
        b.e('s_mul_i32 %s, %s, %s' % (sreg(S_T + 1), sreg(S_T + 1), sreg(S_T)))
        b.e('s_lshl_b32 %s, %s, 14' % (sreg(S_T + 1), sreg(S_T + 1)))                    # 16-KiB sub-tiles (the tiled copy of a layer is < 2^31 bytes: host)
        b.e('s_add_u32 %s, %s, %s' % (sreg(S_W1), sreg(S_W1), sreg(S_T + 1)))
        b.e('s_addc_u32 %s, %s, 0' % (sreg(S_W1 + 1), sreg(S_W1 + 1)))
        b.e('s_lshl_b32 %s, %s, %d' % (sreg(S_T + 1), sreg(S_T), 14 + (4 if hs else 3)))  # NQ = 16 / 8 regressor sub-tiles per column step
        b.e('s_add_u32 %s, %s, %s' % (sreg(S_W2), sreg(S_W2), sreg(S_T + 1)))
        b.e('s_addc_u32 %s, %s, 0' % (sreg(S_W2 + 1), sreg(S_W2 + 1)))
        b.e('s_mul_i32 %s, %s, s3' % (sreg(S_T), sreg(S_N1)))
        b.e('s_lshl_b32 %s, %s, 2' % (sreg(S_T), sreg(S_T)))
        b.e('s_add_u32 %s, %s, %s' % (sreg(S_B1), sreg(S_B1), sreg(S_T)))
        b.e('s_addc_u32 %s, %s, 0' % (sreg(S_B1 + 1), sreg(S_B1 + 1)))
        b.e('s_mul_i32 %s, %s, %s' % (sreg(S_T), sreg(S_M), sreg(S_LDO)))
        b.e('s_lshl_b32 %s, %s, 2' % (sreg(S_T), sreg(S_T)))
        b.e('s_sub_u32 %s, s3, 1' % sreg(S_T + 1))
        b.e('s_mul_hi_u32 %s, %s, %s' % (sreg(S_T + 2), sreg(S_T), sreg(S_T + 1)))
        b.e('s_mul_i32 %s, %s, %s' % (sreg(S_T), sreg(S_T), sreg(S_T + 1)))
        b.e('s_add_u32 %s, %s, %s' % (sreg(S_OUT), sreg(S_PART), sreg(S_T)))
        b.e('s_addc_u32 %s, %s, %s' % (sreg(S_OUT + 1), sreg(S_PART + 1), sreg(S_T + 2)))
        b.label('L_cs_done')
    stamp(b, 0, 0)
    b.e('s_lshr_b32 %s, %s, %d' % (sreg(S_NSUB1), sreg(S_K1), 4 if hs else 5))               # sub-tiles of 32 k (bf16) / 16 k (split-f16)
    b.e('s_lshr_b32 %s, %s, 8' % (sreg(S_NCOL), sreg(S_N1)))
    b.e('s_lshl_b32 %s, %s, 9' % (sreg(S_COLBYTES), sreg(S_LDB1)))          # 256 rows x ldb1 halves x 2 B
    b.e('s_lshl_b32 %s, %s, 12' % (sreg(S_DMA), sreg(S_WAVE)))              # this wave's 4 pieces: image rows 64 w ..
    # ---- bias tables: bias1s[i] = out_scale * bias1[i] (i < N1), bias2s[i] = bias2[i] (i < n2, else 0; 256 entries).
    # One workgroup per CU: nothing overlaps a band's prologue.  So every load of the tables is requested at once (N1 / 256 <= 16 per thread + one
    # for bias2) and, behind them, this wave's LDS-DMA pieces of the first four weight sub-tiles; ONE counted wait (the 16 pieces stay in flight) releases
    # the table values.  (First form: load - wait - write per 256 entries, then the pieces from the role's prologue: five memory round trips in a row.)
    BL = 142                                                                 # v142 .. v157: bias1 values, v158: bias2
    b.e('v_lshlrev_b32_e32 %s, 2, v0' % vreg(V_T + 1))                                   # byte offset of entry tid
    for it in range(MAX_N1 // 256):
        b.e('s_cmp_gt_u32 %s, %d' % (sreg(S_N1), 256 * it))
        b.e('s_cbranch_scc0 L_b1_issued')
        b.e('global_load_dword %s, %s, %s offset:%d' % (vreg(BL + it), vreg(V_T + 1), sreg(S_B1, 2), 1024 * it) if 1024 * it < 4096 else
            'global_load_dword %s, %s, %s' % (vreg(BL + it), vreg(V_T + 2), sreg(S_B1, 2)))
        if 1024 * (it + 1) >= 4096 and it + 1 < MAX_N1 // 256:                           # (13-bit immediates: the byte offset moves on in a register)
            b.e('v_add_u32_e32 %s, %d, %s' % (vreg(V_T + 2), 1024 * (it + 1), vreg(V_T + 1)))
    b.label('L_b1_issued')
    b.e('s_lshl_b32 %s, %s, 2' % (sreg(S_BIAS2OFF), sreg(S_N1)))
    b.e('s_add_u32 %s, %s, %d' % (sreg(S_BIAS2OFF), sreg(S_BIAS2OFF), BIAS1_OFF))
    b.e('v_mov_b32_e32 %s, 0' % vreg(BL + 16))
    b.e('v_cmp_gt_u32_e32 vcc, %s, v0' % sreg(S_N2))
    if 'colsplit' in dbg:                                  # splits 1 .. carry no bias2 (the host adds their outputs to split 0's)
        b.e('s_cmp_eq_u32 s3, 0')
        b.e('s_cselect_b64 %s, -1, 0' % sreg(S_T, 2))
        b.e('s_and_b64 vcc, vcc, %s' % sreg(S_T, 2))
    b.e('s_and_saveexec_b64 %s, vcc' % sreg(S_SAVE, 2))
    b.e('global_load_dword %s, %s, %s' % (vreg(BL + 16), vreg(V_T + 1), sreg(S_B2, 2)))
    b.e('s_mov_b64 exec, %s' % sreg(S_SAVE, 2))
    # the first four sub-tiles of the pair layer's weight stream (the roles' prologues count them as in flight)
    b.e('s_mov_b32 %s, %s' % (sreg(S_W1P), sreg(S_W1)))
    b.e('s_mov_b32 %s, %s' % (sreg(S_W1P + 1), sreg(S_W1 + 1)))
    b.e('v_lshlrev_b32_e32 %s, 4, %s' % (vreg(V_VO), vreg(V_LANE)))
    b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_VO), sreg(S_DMA), vreg(V_VO)))
    npieces = 0
    if 'nodma' not in dbg:
        for t in range(4):
            b.e('s_add_u32 m0, %s, %d' % (sreg(S_DMA), t * RING_SLOT))
            b.e('s_nop 0')
            for pc in range(4):
                b.e('global_load_lds_dwordx4 %s, %s%s' % (vreg(V_VO), sreg(S_W1P, 2), ' offset:%d' % (1024 * pc) if pc else ''))
                npieces += 1
            b.e('s_add_u32 %s, %s, %d' % (sreg(S_W1P), sreg(S_W1P), RING_SLOT))
            b.e('s_addc_u32 %s, %s, 0' % (sreg(S_W1P + 1), sreg(S_W1P + 1)))
    else:
        b.e('s_add_u32 %s, %s, %d' % (sreg(S_W1P), sreg(S_W1P), 4 * RING_SLOT))
        b.e('s_addc_u32 %s, %s, 0' % (sreg(S_W1P + 1), sreg(S_W1P + 1)))
    # ---- lane constants
    b.e('v_and_b32_e32 %s, 31, %s' % (vreg(V_L31), vreg(V_LANE)))
    b.e('v_lshrrev_b32_e32 %s, 5, %s' % (vreg(V_HI), vreg(V_LANE)))
    if hs:
        b.e('v_lshlrev_b32_e32 %s, 1, %s' % (vreg(V_T + 6), vreg(V_HI)))                     # unit 2 hi of the 4 units (16 k) of a slab row
    else:
        b.e('s_lshl_b32 %s, %s, 2' % (sreg(S_T + 4), sreg(S_H)))
        b.e('v_lshl_add_u32 %s, %s, 1, %s' % (vreg(V_T + 6), vreg(V_HI), sreg(S_T + 4)))     # unit 4 h + 2 hi: this half's k-step of a slab row
    # pr0 = m0 / nt and prmax = (M - 1) / nt (uniform; exact division by multiplication + one correction)
    for k, src in ((0, sreg(S_M0)), (1, None)):
        if src is None:
            b.e('s_sub_u32 %s, %s, 1' % (sreg(S_T), sreg(S_M)))
            src = sreg(S_T)
        b.e('v_mov_b32_e32 %s, %s' % (vreg(V_T + 8), src))
        b.e('v_mul_hi_u32 %s, %s, %s' % (vreg(V_T + 9), vreg(V_T + 8), sreg(S_MAGIC)))
        b.e('v_mul_lo_u32 %s, %s, %s' % (vreg(V_T + 10), vreg(V_T + 9), sreg(S_NT)))
        b.e('v_sub_u32_e32 %s, %s, %s' % (vreg(V_T + 10), vreg(V_T + 8), vreg(V_T + 10)))
        b.e('v_cmp_le_u32_e32 vcc, %s, %s' % (sreg(S_NT), vreg(V_T + 10)))
        b.e('v_addc_co_u32_e32 %s, vcc, 0, %s, vcc' % (vreg(V_T + 11 + k), vreg(V_T + 9)))    # q + (r >= nt)
    # ---- this lane's two rows: m = m0 + 64 rp + 32 r + l31 (clamped for the loads), pr = m / nt, t = m - pr nt
    rsh = 6 if hs else 7                                     # log2 of the bytes of a slab row (16 / 32 k fp32)

    def row_of(r_off_sgpr_or_imm, dst_m):
        b.e('s_lshl_b32 %s, %s, 6' % (sreg(S_T), sreg(S_RP)))
        if isinstance(r_off_sgpr_or_imm, int):
            b.e('s_add_u32 %s, %s, %d' % (sreg(S_T), sreg(S_T), r_off_sgpr_or_imm))
        else:
            b.e('s_lshl_b32 %s, %s, 5' % (sreg(S_T + 1), r_off_sgpr_or_imm))
            b.e('s_add_u32 %s, %s, %s' % (sreg(S_T), sreg(S_T), sreg(S_T + 1)))
        b.e('s_add_u32 %s, %s, %s' % (sreg(S_T), sreg(S_T), sreg(S_M0)))
        b.e('v_add_u32_e32 %s, %s, %s' % (vreg(dst_m), sreg(S_T), vreg(V_L31)))

    def slab_addresses(m_reg, k):
        """pr, t of row m (clamped) -> the LDS addresses of this lane's T units and L0 units: V_TL0 + k, V_TL1 + k, V_LL + k"""
        b.e('s_sub_u32 %s, %s, 1' % (sreg(S_T), sreg(S_M)))
        b.e('v_min_u32_e32 %s, %s, %s' % (vreg(V_T), sreg(S_T), vreg(m_reg)))
        b.e('v_mul_hi_u32 %s, %s, %s' % (vreg(V_T + 1), vreg(V_T), sreg(S_MAGIC)))
        b.e('v_mul_lo_u32 %s, %s, %s' % (vreg(V_T + 2), vreg(V_T + 1), sreg(S_NT)))
        b.e('v_sub_u32_e32 %s, %s, %s' % (vreg(V_T + 3), vreg(V_T), vreg(V_T + 2)))
        b.e('v_cmp_le_u32_e32 vcc, %s, %s' % (sreg(S_NT), vreg(V_T + 3)))
        b.e('v_cndmask_b32_e64 %s, 0, 1, vcc' % vreg(V_T + 4))
        b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_T + 1), vreg(V_T + 1), vreg(V_T + 4)))      # pr
        b.e('v_mul_lo_u32 %s, %s, %s' % (vreg(V_T + 4), vreg(V_T + 4), sreg(S_NT)))
        b.e('v_sub_u32_e32 %s, %s, %s' % (vreg(V_T + 3), vreg(V_T + 3), vreg(V_T + 4)))      # t
        # T slab: unit u of row t sits at t * rowbytes + ((u ^ swizzle(t)) << 4); swizzle (t >> 1) & 7 for 8 units, (t >> 2) & 3 for 4
        if hs:
            b.e('v_bfe_u32 %s, %s, 2, 2' % (vreg(V_T + 5), vreg(V_T + 3)))
        else:
            b.e('v_bfe_u32 %s, %s, 1, 3' % (vreg(V_T + 5), vreg(V_T + 3)))
        b.e('v_lshlrev_b32_e32 %s, %d, %s' % (vreg(V_T + 7), rsh, vreg(V_T + 3)))
        b.e('v_add_u32_e32 %s, %d, %s' % (vreg(V_T + 7), TS_OFF, vreg(V_T + 7)))
        b.e('v_xor_b32_e32 %s, %s, %s' % (vreg(V_T + 2), vreg(V_T + 6), vreg(V_T + 5)))
        b.e('v_lshl_add_u32 %s, %s, 4, %s' % (vreg(V_TL0 + k), vreg(V_T + 2), vreg(V_T + 7)))
        b.e('v_or_b32_e32 %s, 1, %s' % (vreg(V_T + 2), vreg(V_T + 6)))
        b.e('v_xor_b32_e32 %s, %s, %s' % (vreg(V_T + 2), vreg(V_T + 2), vreg(V_T + 5)))
        b.e('v_lshl_add_u32 %s, %s, 4, %s' % (vreg(V_TL1 + k), vreg(V_T + 2), vreg(V_T + 7)))
        # L0 slab: row pr - pr0, this lane's two units (contiguous)
        b.e('v_sub_u32_e32 %s, %s, %s' % (vreg(V_T + 2), vreg(V_T + 1), vreg(V_T + 11)))
        b.e('v_lshlrev_b32_e32 %s, %d, %s' % (vreg(V_T + 2), rsh, vreg(V_T + 2)))
        b.e('v_lshl_add_u32 %s, %s, 4, %s' % (vreg(V_T + 2), vreg(V_T + 6), vreg(V_T + 2)))
        b.e('v_add_u32_e32 %s, %d, %s' % (vreg(V_LL + k), L0S_OFF, vreg(V_T + 2)))

    for r in range(2):
        row_of(32 * r, V_M + r)
        b.e('v_cmp_gt_i32_e32 vcc, %s, %s' % (sreg(S_M), vreg(V_M + r)))
        b.e('s_mov_b64 %s, vcc' % sreg(S_ROWMASK1 if r else S_ROWMASK, 2))
        if not hs:
            slab_addresses(V_M + r, r)                       # bf16: a wave converts its k-step of BOTH row groups
        # output addressing of the row
        b.e('s_lshl_b32 %s, %s, 9' % (sreg(S_T), sreg(S_H)))                                 # 128 h floats = 512 h bytes
        b.e('v_mul_lo_u32 %s, %s, %s' % (vreg(V_OUTOFF + r), vreg(V_M + r), sreg(S_LDO)))
        b.e('v_lshlrev_b32_e32 %s, 2, %s' % (vreg(V_OUTOFF + r), vreg(V_OUTOFF + r)))
        b.e('v_lshl_add_u32 %s, %s, 4, %s' % (vreg(V_OUTOFF + r), vreg(V_HI), vreg(V_OUTOFF + r)))
        b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_OUTOFF + r), sreg(S_T), vreg(V_OUTOFF + r)))
    if hs:                                                   # split-f16: a wave converts the whole fragment of row group r = h
        row_of(sreg(S_H), V_T + 13)
        slab_addresses(V_T + 13, 0)
    # L0 slab DMA (wave 0): lane -> row min(pr0 + (lane >> 3), prmax), 16-byte unit lane & 7
    b.e('v_lshrrev_b32_e32 %s, %d, %s' % (vreg(V_T + 8), 2 if hs else 3, vreg(V_LANE)))
    b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_T + 8), vreg(V_T + 8), vreg(V_T + 11)))
    b.e('v_min_u32_e32 %s, %s, %s' % (vreg(V_T + 8), vreg(V_T + 8), vreg(V_T + 12)))
    b.e('v_mul_lo_u32 %s, %s, %s' % (vreg(V_T + 8), vreg(V_T + 8), sreg(S_LDL)))
    b.e('v_and_b32_e32 %s, %d, %s' % (vreg(V_T + 9), 3 if hs else 7, vreg(V_LANE)))
    b.e('v_lshlrev_b32_e32 %s, 2, %s' % (vreg(V_T + 9), vreg(V_T + 9)))
    b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_T + 8), vreg(V_T + 8), vreg(V_T + 9)))
    b.e('v_lshlrev_b32_e32 %s, 2, %s' % (vreg(V_LOFF), vreg(V_T + 8)))
    # T slab DMA: this wave's chunks min(2 w + i, nch - 1) of the nt * 128 bytes (nch = ceil(nt / 8) <= 8 chunks of 1 KiB)
    b.e('s_lshl_b32 %s, %s, %d' % (sreg(S_TSLABB), sreg(S_NT), rsh))
    b.e('s_add_u32 %s, %s, %d' % (sreg(S_T), sreg(S_NT), 15 if hs else 7))
    b.e('s_lshr_b32 %s, %s, %d' % (sreg(S_T), sreg(S_T), 4 if hs else 3))
    b.e('s_sub_u32 %s, %s, 1' % (sreg(S_T), sreg(S_T)))                                  # nch - 1
    b.e('s_lshl_b32 %s, %s, 1' % (sreg(S_T + 1), sreg(S_WAVE)))
    for i, tch in enumerate((S_TCH0, S_TCH1)):
        b.e('s_add_u32 %s, %s, %d' % (sreg(S_T + 2), sreg(S_T + 1), i))
        b.e('s_min_u32 %s, %s, %s' % (sreg(S_T + 2), sreg(S_T + 2), sreg(S_T)))
        b.e('s_lshl_b32 %s, %s, 10' % (sreg(tch), sreg(S_T + 2)))
        b.e('v_lshlrev_b32_e32 %s, 4, %s' % (vreg(V_TOFF + i), vreg(V_LANE)))
        b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_TOFF + i), sreg(tch), vreg(V_TOFF + i)))
    # ---- weight fragment reads: row (128 h + 32 jj + l31) of the image, 16-byte chunk ((2 s + hi) ^ ((l31 >> 2) & 3))
    b.e('v_bfe_u32 %s, %s, 2, 2' % (vreg(V_T), vreg(V_L31)))
    b.e('v_xor_b32_e32 %s, %s, %s' % (vreg(V_T + 1), vreg(V_HI), vreg(V_T)))
    b.e('v_or_b32_e32 %s, 2, %s' % (vreg(V_T + 2), vreg(V_HI)))
    b.e('v_xor_b32_e32 %s, %s, %s' % (vreg(V_T + 2), vreg(V_T + 2), vreg(V_T)))
    b.e('v_lshlrev_b32_e32 %s, 6, %s' % (vreg(V_T + 3), vreg(V_L31)))
    b.e('s_lshl_b32 %s, %s, 13' % (sreg(S_T), sreg(S_H)))
    b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_T + 3), sreg(S_T), vreg(V_T + 3)))
    b.e('v_lshl_add_u32 %s, %s, 4, %s' % (vreg(V_RD0), vreg(V_T + 1), vreg(V_T + 3)))
    b.e('v_lshl_add_u32 %s, %s, 4, %s' % (vreg(V_RD1), vreg(V_T + 2), vreg(V_T + 3)))
    # ---- fragment exchange areas of this row pair: lane-linear 16 B
    b.e('v_lshlrev_b32_e32 %s, 4, %s' % (vreg(V_AX1), vreg(V_LANE)))
    b.e('s_lshl_b32 %s, %s, 13' % (sreg(S_T), sreg(S_RP)))
    b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_AX1), sreg(S_T), vreg(V_AX1)))
    b.e('v_add_u32_e32 %s, %d, %s' % (vreg(V_AX2), AX2_OFF, vreg(V_AX1)))
    b.e('v_add_u32_e32 %s, %d, %s' % (vreg(V_AX1), AX1_OFF, vreg(V_AX1)))
    # (V_VO - byte 4096 w + 16 lane of the pre-tiled sub-tile = of the ring slot - was set with the first pieces above)
    # ---- bias reads
    b.e('v_lshlrev_b32_e32 %s, 4, %s' % (vreg(V_BADDR), vreg(V_HI)))
    b.e('v_add_u32_e32 %s, %d, %s' % (vreg(V_BADDR), BIAS1_OFF, vreg(V_BADDR)))
    b.e('v_lshlrev_b32_e32 %s, 2, %s' % (vreg(V_4HI), vreg(V_HI)))
    b.e('s_lshl_b32 %s, %s, 9' % (sreg(S_T), sreg(S_H)))
    b.e('v_lshl_add_u32 %s, %s, 4, %s' % (vreg(V_B2ADDR), vreg(V_HI), sreg(S_BIAS2OFF)))
    b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_B2ADDR), sreg(S_T), vreg(V_B2ADDR)))
    # ---- the first four T / L0 slabs (every wave its two 1-KiB chunks of a T slab, wave 0 the band's L0 rows), then the table values: they were requested
    # first and loads return in order, so one counted wait that leaves this wave's LDS-DMA in flight (16 weight pieces + 8 slab chunks; wave 0 has 4 more) releases them
    nslab = 0
    b.e('s_mov_b32 %s, %s' % (sreg(S_L0P), sreg(S_L0)))
    b.e('s_mov_b32 %s, %s' % (sreg(S_L0P + 1), sreg(S_L0 + 1)))
    b.e('s_mov_b32 %s, %s' % (sreg(S_TSP), sreg(S_TS)))
    b.e('s_mov_b32 %s, %s' % (sreg(S_TSP + 1), sreg(S_TS + 1)))
    for t in range(0 if 'roleslabs' in dbg else 4):
        if 'noreq' not in dbg:
            for i, tch in enumerate((S_TCH0, S_TCH1)):
                b.e('s_add_u32 m0, %s, %d' % (sreg(tch), TS_OFF + t * TSLAB))
                b.e('s_nop 0')
                b.e('global_load_lds_dwordx4 %s, %s' % (vreg(V_TOFF + i), sreg(S_TSP, 2)))
                nslab += 1
            b.e('s_cmp_lg_u32 %s, 0' % sreg(S_WAVE))
            b.e('s_cbranch_scc1 L_pro_l0skip_%d' % t)
            b.e('s_mov_b32 m0, %d' % (L0S_OFF + t * 1024))
            b.e('s_nop 0')
            b.e('global_load_lds_dwordx4 %s, %s' % (vreg(V_LOFF), sreg(S_L0P, 2)))
            b.label('L_pro_l0skip_%d' % t)
        b.e('s_add_u32 %s, %s, %s' % (sreg(S_TSP), sreg(S_TSP), sreg(S_TSLABB)))
        b.e('s_addc_u32 %s, %s, 0' % (sreg(S_TSP + 1), sreg(S_TSP + 1)))
        b.e('s_add_u32 %s, %s, %d' % (sreg(S_L0P), sreg(S_L0P), 64 if hs else 128))
        b.e('s_addc_u32 %s, %s, 0' % (sreg(S_L0P + 1), sreg(S_L0P + 1)))
    b.e('s_waitcnt vmcnt(%d)' % (npieces + nslab))                                                 # the table values are in (loads return in order); the pieces fly on
    b.e('v_lshlrev_b32_e32 %s, 2, v0' % vreg(V_T + 1))                                   # (byte offset of entry tid again: the scratch registers were reused)
    b.e('v_add_u32_e32 %s, %d, %s' % (vreg(V_T + 2), BIAS1_OFF, vreg(V_T + 1)))
    for it in range(MAX_N1 // 256):
        b.e('s_cmp_gt_u32 %s, %d' % (sreg(S_N1), 256 * it))
        b.e('s_cbranch_scc0 L_b1_written')
        if hs:                                                                           # h2 is carried as out_scale * relu(z1): the bias in the same scale
            b.e('v_mul_f32_e32 %s, %s, %s' % (vreg(BL + it), sreg(S_OUTSC), vreg(BL + it)))
        b.e('ds_write_b32 %s, %s offset:%d' % (vreg(V_T + 2), vreg(BL + it), 1024 * it))
    b.label('L_b1_written')
    b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_T + 2), sreg(S_BIAS2OFF), vreg(V_T + 1)))
    b.e('ds_write_b32 %s, %s' % (vreg(V_T + 2), vreg(BL + 16)))                          # 256 threads = the 256 entries
    if hs:
        b.e('v_mov_b32_e32 %s, 0' % vreg(V_PK1))
        b.e('v_mov_b32_e32 %s, 0' % vreg(V_PK2))
    for i in range(128):
        b.e('v_accvgpr_write_b32 %s, 0' % areg(ACC2 + i))
    b.e('s_waitcnt lgkmcnt(0)')
    stamp(b, 1, 1)
    b.e('s_cmp_eq_u32 %s, 0' % sreg(S_H))
    b.e('s_cbranch_scc0 L_r1_start')


END_LABEL = ['L_end']           # where a finished band goes: 'L_next' in the persistent form (set by kernel())


def epilogue_staged(b, h, slow_label, nwaves=4):
    """Round 6.  The band's output is ONE contiguous block of 128 x ldo floats when ldo == n2 (the library's layout) - but a lane owns a ROW,
    so the direct stores of the accumulators are 64 scattered 8-byte pieces per instruction: 0.46 ms of the 3.33 ms launch at configs[2]
    (csi_band4_bf16_nostore, profiles/r06_bf16_band_probe.txt).  Here the block is assembled in LDS (the ring, the exchange areas and the
    tables are dead by now) and leaves as a linear stream of 16-byte pieces, 1 KiB per wave-instruction.  Taken when the band is full,
    ldo == n2, n2 is even (8-byte LDS writes) and the block is 16-byte aligned; anything else branches to the row-per-lane stores."""
    ST, D, GOFF = V_T + 12, 32, 60                       # v30, v31: LDS row addresses [r]; v32 ..: bias quads, then the copy's data; v60 ..: its offsets
    BQ2 = 32
    b.e('s_add_u32 %s, %s, 128' % (sreg(S_T), sreg(S_M0)))
    b.e('s_cmp_le_u32 %s, %s' % (sreg(S_T), sreg(S_M)))
    b.e('s_cbranch_scc0 %s' % slow_label)
    b.e('s_cmp_eq_u32 %s, %s' % (sreg(S_LDO), sreg(S_N2)))
    b.e('s_cbranch_scc0 %s' % slow_label)
    b.e('s_and_b32 %s, %s, 1' % (sreg(S_T), sreg(S_N2)))
    b.e('s_cmp_eq_u32 %s, 0' % sreg(S_T))
    b.e('s_cbranch_scc0 %s' % slow_label)
    b.e('s_mul_i32 %s, %s, %s' % (sreg(S_T + 4), sreg(S_M0), sreg(S_LDO)))
    b.e('s_lshl_b32 %s, %s, 2' % (sreg(S_T + 4), sreg(S_T + 4)))                       # (M * ldo * 4 < 2^32: band8_serves)
    b.e('s_add_u32 %s, %s, %s' % (sreg(S_T + 4), sreg(S_OUT), sreg(S_T + 4)))
    b.e('s_addc_u32 %s, %s, 0' % (sreg(S_T + 5), sreg(S_OUT + 1)))
    b.e('s_and_b32 %s, %s, 15' % (sreg(S_T), sreg(S_T + 4)))
    b.e('s_cmp_eq_u32 %s, 0' % sreg(S_T))
    b.e('s_cbranch_scc0 %s' % slow_label)
    # bias2 quads of this half's 16 column chunks
    for jj in range(4):
        for rq in range(4):
            b.e('ds_read_b128 %s, %s offset:%d' % (vreg(BQ2 + 4 * (4 * jj + rq), 4), vreg(V_B2ADDR), (32 * jj + 8 * rq) * 4))
    # LDS row addresses: ((64 rp + 32 r + l31) * ldo + 128 h + 4 hi) * 4
    for r in range(2):
        b.e('s_lshl_b32 %s, %s, 6' % (sreg(S_T), sreg(S_RP)))
        b.e('v_add_u32_e32 %s, %s, %s' % (vreg(ST + r), sreg(S_T), vreg(V_L31)))
        if r:
            b.e('v_add_u32_e32 %s, 32, %s' % (vreg(ST + r), vreg(ST + r)))
        b.e('v_mul_lo_u32 %s, %s, %s' % (vreg(ST + r), vreg(ST + r), sreg(S_LDO)))
        b.e('v_add_u32_e32 %s, %s, %s' % (vreg(ST + r), vreg(ST + r), vreg(V_4HI)))
        b.e('v_add_u32_e32 %s, %d, %s' % (vreg(ST + r), 128 * h, vreg(ST + r)))
        b.e('v_lshlrev_b32_e32 %s, 2, %s' % (vreg(ST + r), vreg(ST + r)))
    b.e('s_waitcnt lgkmcnt(0)')
    b.e('s_barrier')                          # every wave has left the main loop and holds its bias quads: the LDS is the staging area now
    for r in range(2):
        for jj in range(4):
            for rq in range(4):
                c0 = 128 * h + 32 * jj + 8 * rq
                imm = (32 * jj + 8 * rq) * 4
                uid = 'L_s%d_%d_%d_%d' % (h, r, jj, rq)
                bq = BQ2 + 4 * (4 * jj + rq)
                for e in range(4):
                    b.e('v_accvgpr_read_b32 %s, %s' % (vreg(V_T + e), areg(ACC2 + 16 * (4 * r + jj) + 4 * rq + e)))
                for e in range(4):
                    b.e('v_fma_f32 %s, %s, %s, %s' % (vreg(V_T + e), vreg(V_T + e), sreg(S_AS2), vreg(bq + e)))
                b.e('s_cmp_ge_u32 %s, %d' % (sreg(S_N2), c0 + 8))
                b.e('s_cbranch_scc0 %s_m' % uid)
                b.e('ds_write_b64 %s, %s offset:%d' % (vreg(ST + r), vreg(V_T, 2), imm))
                b.e('ds_write_b64 %s, %s offset:%d' % (vreg(ST + r), vreg(V_T + 2, 2), imm + 8))
                b.e('s_branch %s_d' % uid)
                b.label('%s_m' % uid)
                for pr in range(2):                                                     # column pair valid <=> 4 hi + 2 pr < n2 - c0 (n2 even)
                    b.e('s_sub_i32 %s, %s, %d' % (sreg(S_T), sreg(S_N2), c0 + 2 * pr))
                    b.e('v_cmp_gt_i32_e32 vcc, %s, %s' % (sreg(S_T), vreg(V_4HI)))
                    b.e('s_mov_b64 exec, vcc')
                    b.e('ds_write_b64 %s, %s offset:%d' % (vreg(ST + r), vreg(V_T + 2 * pr, 2), imm + 8 * pr))
                b.e('s_mov_b64 exec, -1')
                b.label('%s_d' % uid)
    b.e('s_waitcnt lgkmcnt(0)')
    b.e('s_barrier')
    # ---- the block leaves: piece i (16 bytes) of 32 * ldo, wave w takes pieces 64 w + lane + (64 nwaves) j
    U = 6
    b.e('s_lshl_b32 %s, %s, 6' % (sreg(S_T), sreg(S_WAVE)))
    b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_T + 8), sreg(S_T), vreg(V_LANE)))          # piece index of round 0
    b.e('v_lshlrev_b32_e32 %s, 4, %s' % (vreg(V_T + 9), vreg(V_T + 8)))                 # its LDS address; advances by U rounds per iteration
    for k in range(U):
        b.e('v_add_u32_e32 %s, %d, %s' % (vreg(GOFF + k), 1024 * nwaves * k, vreg(V_T + 9)))   # global offsets of the U rounds (the base advances)
    b.e('s_lshl_b32 %s, %s, 5' % (sreg(S_T + 2), sreg(S_LDO)))                          # pieces
    b.e('s_mov_b32 %s, 0' % sreg(S_T + 3))
    b.label('L_copy_%d' % h)
    for k in range(U):
        b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_T + 10), sreg(S_T + 3), vreg(V_T + 8)))
        if k:
            b.e('v_add_u32_e32 %s, %d, %s' % (vreg(V_T + 10), 64 * nwaves * k, vreg(V_T + 10)))
        b.e('v_cmp_gt_u32_e32 vcc, %s, %s' % (sreg(S_T + 2), vreg(V_T + 10)))
        b.e('s_mov_b64 %s, vcc' % sreg(78 + 2 * k, 2))
        b.e('s_mov_b64 exec, vcc')
        b.e('ds_read_b128 %s, %s offset:%d' % (vreg(D + 4 * k, 4), vreg(V_T + 9), 1024 * nwaves * k))
    b.e('s_mov_b64 exec, -1')
    b.e('s_waitcnt lgkmcnt(0)')
    for k in range(U):
        b.e('s_mov_b64 exec, %s' % sreg(78 + 2 * k, 2))
        b.e('global_store_dwordx4 %s, %s, %s' % (vreg(GOFF + k), vreg(D + 4 * k, 4), sreg(S_T + 4, 2)))
    b.e('s_mov_b64 exec, -1')
    b.e('v_add_u32_e32 %s, %d, %s' % (vreg(V_T + 9), 1024 * nwaves * U, vreg(V_T + 9)))
    b.e('s_add_u32 %s, %s, %d' % (sreg(S_T + 4), sreg(S_T + 4), 1024 * nwaves * U))
    b.e('s_addc_u32 %s, %s, 0' % (sreg(S_T + 5), sreg(S_T + 5)))
    b.e('s_add_u32 %s, %s, %d' % (sreg(S_T + 3), sreg(S_T + 3), 64 * nwaves * U))
    b.e('s_cmp_lt_u32 %s, %s' % (sreg(S_T + 3), sreg(S_T + 2)))
    b.e('s_cbranch_scc1 L_copy_%d' % h)
    b.e('s_branch %s' % END_LABEL[0])


def epilogue(b, h, dbg=(), hs=False):
    b.label('L_epilogue_%d' % h)
    b.e('s_waitcnt vmcnt(0)')                 # the re-fetched head of the stream has landed: the ring may go
    b.e('s_nop 15')
    b.e('s_nop 15')
    stamp(b, 2, 10 + h)
    if hs:
        # range guard (gemm_hs.hip.h hs_report_peak): pk1 = this lane's maximum of the |h1| hi halves, pk2 = of the |h2| hi halves
        b.e('s_cmp_eq_u64 %s, 0' % sreg(S_PEAK, 2))
        b.e('s_cbranch_scc1 L_noguard_%d' % h)
        for pk, row_max in ((V_PK1, True), (V_PK2, False)):
            b.e('v_lshrrev_b32_e32 %s, 16, %s' % (vreg(V_T), vreg(pk)))
            b.e('v_and_b32_e32 %s, 0xffff, %s' % (vreg(V_T + 1), vreg(pk)))
            b.e('v_max_u32_e32 %s, %s, %s' % (vreg(V_T), vreg(V_T), vreg(V_T + 1)))
            b.e('v_min_u32_e32 %s, 0x7c00, %s' % (vreg(V_T), vreg(V_T)))
            b.e('v_cvt_f32_f16_e32 %s, %s' % (vreg(V_T + 1), vreg(V_T)))
            b.e('v_mov_b32_e32 %s, 0' % vreg(V_T + 2))
            b.e('v_cmp_lt_f32_e32 vcc, 0x476a6000, %s' % vreg(V_T + 1))                 # 60000.0 < m
            b.e('s_and_saveexec_b64 %s, vcc' % sreg(S_SAVE, 2))
            b.e('global_atomic_umax %s, %s, %s' % (vreg(V_T + 2), vreg(V_T + 1), sreg(S_PEAK, 2)))
            b.e('s_mov_b64 exec, %s' % sreg(S_SAVE, 2))
            if row_max:
                b.e('v_cmp_lt_f32_e32 vcc, 0, %s' % vreg(V_T + 1))
                b.e('s_and_saveexec_b64 %s, vcc' % sreg(S_SAVE, 2))
                b.e('v_cmp_gt_f32_e32 vcc, 0x3d800000, %s' % vreg(V_T + 1))              # m < 0.0625
                b.e('s_and_b64 exec, exec, vcc')
                b.e('v_mov_b32_e32 %s, 1' % vreg(V_T + 3))
                b.e('global_atomic_or %s, %s, %s offset:4' % (vreg(V_T + 2), vreg(V_T + 3), sreg(S_PEAK, 2)))
                b.e('s_mov_b64 exec, %s' % sreg(S_SAVE, 2))
        b.label('L_noguard_%d' % h)
    if 'nostore' in dbg:
        b.e('s_branch %s' % END_LABEL[0])
    if 'rowstores' not in dbg:
        epilogue_staged(b, h, 'L_rowstores_%d' % h)
    b.label('L_rowstores_%d' % h)
    # ---- output, row per lane: register quad = 4 consecutive outputs (columns 128 h + 32 jj + 8 rq + 4 hi + e)
    for r in range(2):
        mask = sreg(S_ROWMASK1 if r else S_ROWMASK, 2)
        b.e('s_mov_b64 exec, %s' % mask)
        for jj in range(4):
            for rq in range(4):
                c0 = 128 * h + 32 * jj + 8 * rq
                imm = (32 * jj + 8 * rq) * 4
                uid = 'L_o%d_%d_%d_%d' % (h, r, jj, rq)
                b.e('ds_read_b128 %s, %s offset:%d' % (vreg(V_T + 4, 4), vreg(V_B2ADDR), imm))
                for e in range(4):
                    b.e('v_accvgpr_read_b32 %s, %s' % (vreg(V_T + e), areg(ACC2 + 16 * (4 * r + jj) + 4 * rq + e)))
                b.e('s_waitcnt lgkmcnt(0)')
                for e in range(4):
                    b.e('v_fma_f32 %s, %s, %s, %s' % (vreg(V_T + e), vreg(V_T + e), sreg(S_AS2), vreg(V_T + 4 + e)))
                b.e('s_cmp_ge_u32 %s, %d' % (sreg(S_N2), c0 + 8))
                b.e('s_cbranch_scc0 %s_m' % uid)
                b.e('global_store_dwordx2 %s, %s, %s offset:%d' % (vreg(V_OUTOFF + r), vreg(V_T, 2), sreg(S_OUT, 2), imm))
                b.e('global_store_dwordx2 %s, %s, %s offset:%d' % (vreg(V_OUTOFF + r), vreg(V_T + 2, 2), sreg(S_OUT, 2), imm + 8))
                b.e('s_branch %s_d' % uid)
                b.label('%s_m' % uid)
                for e in range(4):
                    b.e('s_sub_i32 %s, %s, %d' % (sreg(S_T), sreg(S_N2), c0 + e))          # column valid <=> 4 hi < n2 - c0 - e
                    b.e('v_cmp_gt_i32_e32 vcc, %s, %s' % (sreg(S_T), vreg(V_4HI)))
                    b.e('s_and_b64 exec, vcc, %s' % mask)
                    b.e('global_store_dword %s, %s, %s offset:%d' % (vreg(V_OUTOFF + r), vreg(V_T + e), sreg(S_OUT, 2), imm + 4 * e))
                b.e('s_mov_b64 exec, %s' % mask)
                b.label('%s_d' % uid)
    b.e('s_mov_b64 exec, -1')
    stamp(b, 3, 20 + h)
    b.e('s_branch %s' % END_LABEL[0])


def kernel(name, dbg=()):
    hs = 'hs' in dbg
    out = ['.globl %s' % name, '.p2align 8', '.type %s,@function' % name, '%s:' % name]
    END_LABEL[0] = 'L_next' if 'persist' in dbg else 'L_end'
    pre = Block('common')
    common_prologue(pre, dbg, hs)
    blocks = [pre]
    role = RoleH if hs else Role4
    r0 = role(0, dbg).build()
    r1 = role(1, dbg).build()
    e0, e1 = Block('ep0'), Block('ep1')
    epilogue(e0, 0, dbg, hs)
    epilogue(e1, 1, dbg, hs)
    blocks += r0 + [e0] + r1 + [e1]
    end = Block('end')
    if 'persist' in dbg:
        end.label('L_next')
        end.e('s_waitcnt lgkmcnt(0)')       # every wave's reads of the staged output block are back before any wave's next prologue writes LDS
        end.e('s_barrier')
        end.e('s_add_u32 s2, s2, %s' % sreg(S_PART))
        end.e('s_branch L_restart')
    end.label('L_end')
    end.e('s_endpgm')
    blocks.append(end)
    for b in blocks:
        for it in b.items:
            if isinstance(it, tuple):
                continue
            out.append(it.text() if isinstance(it, Wait) else it)
    text = '\n'.join(out)
    return re.sub(r'\bL_\w+', lambda m: name + '_' + m.group(0), text)      # labels are per kernel


DESCRIPTOR4 = DESCRIPTOR.replace('.amdhsa_next_free_vgpr 256', '.amdhsa_next_free_vgpr 512').replace('.amdhsa_accum_offset 128', '.amdhsa_accum_offset 256')
META4 = META_KERNEL.replace('.vgpr_count: 256', '.vgpr_count: 512').replace('.agpr_count: 128', '.agpr_count: 256').replace('.max_flat_workgroup_size: 512', '.max_flat_workgroup_size: 256')

VARIANTS = [('csi_band4', ('hs',)), ('csi_band4_p', ('hs', 'persist')), ('csi_band4_bf16_p', ('persist',)), ('csi_band4_cs', ('hs', 'colsplit')), ('csi_band4_bf16_cs', ('colsplit',)), ('csi_band4_rowstores', ('hs', 'rowstores')), ('csi_band4_roleslabs', ('hs', 'roleslabs')), ('csi_band4_bf16_roleslabs', ('roleslabs',)), ('csi_band4_skeleton', ('hs', 'noconv', 'noreq', 'nodma', 'noread')), ('csi_band4_skeleton_nobarrier', ('hs', 'noconv', 'noreq', 'nodma', 'noread', 'nobarrier')), ('csi_band4_noaside', ('hs', 'noconv', 'noreq')),
            ('csi_band4_nodma', ('hs', 'nodma')), ('csi_band4_nostore', ('hs', 'nostore')), ('csi_band4_noconv', ('hs', 'noconv')),
            ('csi_band4_bf16', ()), ('csi_band4_bf16_noconv', ('noconv',)), ('csi_band4_bf16_noaside', ('noconv', 'noreq')),
            ('csi_band4_bf16_skeleton', ('noconv', 'noreq', 'nodma', 'noread')), ('csi_band4_bf16_nodma', ('nodma',)), ('csi_band4_bf16_noread', ('noread',)),
            ('csi_band4_bf16_nobarrier', ('nobarrier',)), ('csi_band4_bf16_nointerleave', ('nointerleave',)),
            ('csi_band4_bf16_noaside_nodma', ('noconv', 'noreq', 'nodma')), ('csi_band4_bf16_noaside_noread', ('noconv', 'noreq', 'noread')),
            ('csi_band4_bf16_skeleton_nobarrier', ('noconv', 'noreq', 'nodma', 'noread', 'nobarrier')),
            ('csi_band4_bf16_pk', ('pk',)), ('csi_band4_bf16_rowstores', ('rowstores',)), 
            ('csi_band4_bf16_nostore', ('nostore',)), ('csi_band4_bf16_skeleton_nostore', ('noconv', 'noreq', 'nodma', 'noread', 'nostore'))]


def parts(only=None):
    text, meta = [], []
    for name, dbg in VARIANTS:
        if only and name not in only:
            continue
        cs = 'colsplit' in dbg
        karg = KARG_BYTES_CS if (cs or 'persist' in dbg) else KARG_BYTES
        text.append(kernel(name, dbg))
        text.append(DESCRIPTOR4.format(name=name, karg=karg, lds=LDS_BYTES, idy=1 if cs else 0))
        meta.append(META4.format(name=name, karg=karg, lds=LDS_BYTES))
    return text, meta


def main():
    """one code object with the requested kernels of BOTH generators (the library loads a single module): csi_band8* names come from
    band_kernel_gen.py, csi_band4* from this file; no names = every variant of both"""
    import band_kernel_gen as g8
    path = sys.argv[1] if len(sys.argv) > 1 else 'band_gfx950.s'
    only = sys.argv[2:] or None
    text, meta = [], []
    for name, dbg in g8.VARIANTS:
        if only and name not in only:
            continue
        cs = 'colsplit' in dbg
        karg = g8.KARG_BYTES_CS if cs else g8.KARG_BYTES
        text.append(g8.kernel(name, dbg, 'bf16' if 'bf16' in dbg else 'hs'))
        text.append(g8.DESCRIPTOR.format(name=name, karg=karg, lds=g8.LDS_BYTES, idy=1 if cs else 0))
        meta.append(g8.META_KERNEL.format(name=name, karg=karg, lds=g8.LDS_BYTES))
    t4, m4 = parts(only)
    out = ['.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', '.text'] + text + t4
    out.append('.amdgpu_metadata\n---\namdhsa.version: [1, 2]\namdhsa.target: amdgcn-amd-amdhsa--gfx950\namdhsa.kernels:\n' + ''.join(meta + m4) + '...\n.end_amdgpu_metadata\n')
    with open(path, 'w') as f:
        f.write('\n'.join(out))


if __name__ == '__main__':
    main()
