#!/usr/bin/env python3
"""band_kernel_gen.py - writes the gfx950 assembly of the fused pair-layer + regressor kernel ("band8").

What the kernel computes (massiveMIMO_CSI_prediction_DNN.py:211-227, the shipped 2-hidden-layer network on the shared
layer-0 path; operands as csi_load_weights prepares them for the split-f16 engine, gemm_hs.hip.h):

    h1[m][k]   = relu(in_scale * L0[m / nt][k] + Ts[m % nt][k])                        (bn0 folded into W1 / bias1)
    h2[m][n]   = out_scale * relu(acc_scale1 * sum_k split(h1)[m][k] * W1[n][k] + bias1[n])      (bn1 folded into W2 / bias2)
    out[m][o]  = acc_scale2 * sum_n split(h2)[m][n] * W2[o][n] + bias2[o]

with every product a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_f16.  h2 never leaves the registers.

Why assembly.  The C++ form of the same idea (gemm_hs_band.hip.h, 4 waves x 512 registers) is correct but (a) hipcc
copies / spills whole accumulator sets once both fill the AGPR file, and (b) measured (profiles/r03_band_probe_4wave.txt):
a wave that is ALONE on its SIMD pays for every VALU instruction in matrix-pipe time - the MFMA + weight-stream skeleton
runs 1.19 ms per 262144 rows, the full kernel 1.9-2.0, the difference being the operand conversion.  So this kernel keeps
TWO waves per SIMD that share their 32 activation rows and split the output features:

  workgroup = 8 waves = one band of 128 pair rows.  wave w: row group rg = w & 3 (rows 32 rg .. +31), half h = w >> 2.
  Waves w and w + 4 sit on the same SIMD.  In a column step of 256 features half h owns feature tiles 4h .. 4h+3 of stage 1
  (64 accumulator registers) and output tiles 4h .. 4h+3 of stage 2 (64 more, alive across the column steps): 128 AGPRs,
  and < 128 VGPRs for fragments, look-ahead values and addresses - 256 registers per wave, two waves per SIMD.

  MFMA operands are swapped (A = weight fragment, B = activation fragment), so lane = activation row and a lane's 8
  accumulator registers of 16 consecutive features are - after bias / relu / split - the B operand of the regressor
  product in the k order {0-3, 8-11 | 4-7, 12-15}; the regressor weights are stored in that order (hs_band_kperm).
  The activation fragment of a sub-step (32 rows x 16 k, hi + lo = 8 registers) is needed by BOTH waves of a pair: one
  of them produces it (VALU, while the partner's MFMAs keep the SIMD's matrix pipe busy), leaves a copy in a 2 KiB LDS
  slot, and the partner reads it behind the sub-step's barrier.  Producers alternate: stage-1 fragment u by half u & 1,
  stage-2 fragment q by the half that owns its accumulator tile.

  The LDS carries the weight stream (ring of 4 sub-tiles of 16 KiB = [256 rows][16 k as hi | lo], 4 sub-steps ahead,
  2 LDS-DMA pieces per wave and sub-step), the fragment exchange (16 KiB) and the two bias tables.

Every vector-memory wait is a counted s_waitcnt vmcnt(N); N comes from a simulation of each role's in-order queue over
every control-flow path (the minimum over the paths is taken, which is always safe).

Usage: band_kernel_gen.py out.s            (then: clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c; ld.lld -shared)
"""
import re
import sys

# ----------------------------------------------------------------------------------------------- layout constants
RING_SLOT = 16384
AX_OFF = 4 * RING_SLOT            # fragment exchange: [rg][parity][hi | lo][64 lanes x 16 B]
BIAS1_OFF = AX_OFF + 16384        # out_scale * bias1 [N1], then bias2 [256]
KARG_BYTES = 128
KARG_BYTES_CS = 144                # column-split form: + the base of the partial outputs (8 bytes, padded to 16)
MAX_N1 = 4096                     # static LDS: ring + exchange + bias tables for up to this many hidden features
LDS_OLD = BIAS1_OFF + 4 * MAX_N1 + 1024
# bf16 'staged' form: the L0 / T values of the stage-1 fragments come through LDS as well - per sub-step one slab of the pilot table
# ([nt rows][32 k] fp32, 16-byte units XOR-swizzled by (row >> 1) & 7, laid out that way in HBM by band_tsw_kernel so that the
# LDS-DMA is linear) and one of the band's L0 rows ([<= 8 rows][32 k]) - rings of 4 slabs, 5 sub-steps ahead
TS_OFF = LDS_OLD                  # 4 x 8 KiB (nt <= 64)
TSLAB = 8192
L0S_OFF = TS_OFF + 4 * TSLAB      # 4 x 1 KiB
LDS_BYTES = L0S_OFF + 4 * 1024

# scalar registers
S_L0, S_TS, S_W1, S_B1, S_W2, S_B2, S_OUT, S_PEAK = 4, 6, 8, 10, 12, 14, 16, 18
S_STAMPS, S_LDL, S_NT, S_LDB1, S_M, S_K1, S_N1, S_LDB2, S_N2, S_LDO = 20, 22, 23, 24, 25, 26, 27, 28, 29, 30
S_INSC, S_AS1OS, S_OUTSC, S_AS2, S_MAGIC = 31, 32, 33, 34, 35
S_WAVE, S_RG, S_H, S_M0 = 36, 37, 38, 39
S_W1P, S_W2P, S_L0P, S_TSP = 40, 42, 44, 46
S_TRIP, S_COL, S_NCOL, S_NSUB1, S_DMA = 48, 49, 50, 51, 52
S_T = 54                          # s54..s63 scratch
S_ROWMASK, S_SAVE = 64, 66
S_COLBYTES, S_BIAS2OFF, S_DMA2 = 68, 69, 70
S_PART = 74                       # column-split form: s[74:75] = base of the partial outputs of splits 1 ..
S_TSLABB, S_TCH = 72, 73        # staged bf16 form: bytes of a T slab (nt * 128), byte offset of this wave's 1-KiB chunk of it

# vector registers (arch half: v0..v127)
V_TID, V_LANE = 0, 1
V_WHI, V_WLO = 2, 18
V_AHI, V_ALO = (34, 42), (38, 46)
V_LV, V_TV, V_GV, V_BQ = 50, 58, 66, 74
V_RDHI, V_RDLO, V_AX, V_LOFF, V_TOFF, V_PK1, V_PK2, V_BADDR = 82, 83, 84, 85, 86, 91, 92, 93
V_VO1, V_VO2 = 117, 121           # LDS-DMA offsets: [own piece 0, own piece 1, partner's piece 0, partner's piece 1]
V_T = 94                          # v94..v109 scratch (prologue, epilogue, stamps); bf16 mode: also the 16 requested Ts values
V_TV16 = 94
V_TL0, V_TL1, V_LL = 87, 88, 89    # staged bf16 form: LDS addresses of this lane's two T units and of its L0 units; V_LOFF / V_TOFF hold the DMA lane offsets
V_OUTOFF, V_M, V_4HI, V_B2ADDR, V_HI, V_L31 = 110, 112, 113, 114, 115, 116
ACC1, ACC2 = 0, 64                # AGPR bases


def sreg(i, n=1):
    return 's%d' % i if n == 1 else 's[%d:%d]' % (i, i + n - 1)


def vreg(i, n=1):
    return 'v%d' % i if n == 1 else 'v[%d:%d]' % (i, i + n - 1)


def areg(i, n=1):
    return 'a%d' % i if n == 1 else 'a[%d:%d]' % (i, i + n - 1)


class Wait:
    """placeholder of a counted vmcnt wait; `needs` = tags whose last issued operation must have completed"""

    def __init__(self, needs):
        self.needs = set(needs)
        self.n = None              # min over the simulated paths

    def text(self):
        return '  s_waitcnt vmcnt(%d)' % min(self.n if self.n is not None else 0, 63)


class Block:
    def __init__(self, name):
        self.name = name
        self.items = []            # str | Wait | ('vm', tag)

    def e(self, s):
        self.items.append('  ' + s)

    def label(self, s):
        self.items.append(s + ':')

    def vm(self, s, tag):
        self.items.append('  ' + s)
        self.items.append(('vm', tag))

    def vm_opt(self, s, tag):
        """an operation only SOME waves of the role issue (behind a scalar branch): counted waits take the minimum of both cases"""
        self.items.append('  ' + s)
        self.items.append(('vmopt', tag))

    def wait_vm(self, needs):
        w = Wait(needs)
        self.items.append(w)
        return w


def simulate(blocks_in_order, queue=None, with_opt=True):
    """walk the blocks in execution order; every Wait gets n = min(n, operations younger than its newest needed one)"""
    q = list(queue or [])
    for b in blocks_in_order:
        for it in b.items:
            if isinstance(it, tuple):
                if it[0] == 'unit' or (it[0] == 'vmopt' and not with_opt):
                    continue
                q.append(it[1])
            elif isinstance(it, Wait):
                pos = -1
                for i, t in enumerate(q):
                    if t in it.needs:
                        pos = i
                if pos >= 0:
                    n = len(q) - 1 - pos
                    it.n = n if it.n is None else min(it.n, n)
                    q = q[pos + 1:]
                elif it.n is None:
                    it.n = 63      # nothing of it in flight on this path
    return q


class Role:
    """straight-line program of one wave half (h = 0 / 1)"""

    def __init__(self, h, dbg, mode='hs'):
        self.h = h
        self.dbg = dbg
        self.mode = mode
        self.bf = mode == 'bf16'
        self.nq = 8 if self.bf else 16          # stage-2 sub-steps per column step (sub-tiles of 32 / 16 k)
        self.tv = V_TV16 if self.bf else V_TV
        self.staged = 'nostage' not in dbg
        self.slab_k = 32 if self.bf else 16       # k-columns of a T / L0 slab = of a sub-step
        self.uid = 0

    # ---------------------------------------------------------------- pieces of code
    def mfma(self, b, acc, w, a, zero_c=False):
        b.e('v_mfma_f32_32x32x16_%s %s, %s, %s, %s' % ('bf16' if self.bf else 'f16', areg(acc, 16), vreg(w, 4), vreg(a, 4), '0' if zero_c else areg(acc, 16)))

    def read_w(self, b, plane, slot):
        base, addr = (V_WLO, V_RDLO) if plane else (V_WHI, V_RDHI)
        if 'noread' in self.dbg:
            return
        for jj in range(4):
            b.e('ds_read_b128 %s, %s offset:%d' % (vreg(base + 4 * jj, 4), vreg(addr), slot * RING_SLOT + jj * 2048))

    def request(self, b, mode, xset=0):
        """L0 / Ts values of a stage-1 fragment.  hs: this role's next fragment (16 k = 2 x 16 B per lane and array), requested by
        the half that will convert it.  bf16: BOTH halves convert every fragment, each its own 16-k MFMA step of the 32 (half 0:
        k 8 hi .. +7, half 1: 16 + 8 hi ..), so each requests 2 x 16 B per array into register set `xset` (fragment parity).
        mode 'reset': first of a column."""
        if self.staged:
            return
        if mode == 'reset':
            for p, src in ((S_L0P, S_L0), (S_TSP, S_TS)):
                b.e('s_add_u32 %s, %s, %d' % (sreg(p), sreg(src), 0 if self.bf else 64 * self.h))
                b.e('s_addc_u32 %s, %s, 0' % (sreg(p + 1), sreg(src + 1)))
        if 'noreq' not in self.dbg:
            offs = ((0, 16), (64, 80))[self.h] if self.bf else (0, 16)
            base = 8 * xset if self.bf else 0
            for dst, voff, ptr in ((V_LV, V_LOFF, S_L0P), (self.tv, V_TOFF, S_TSP)):
                for i, o in enumerate(offs):
                    b.vm('global_load_dwordx4 %s, %s, %s%s' % (vreg(dst + base + 4 * i, 4), vreg(voff), sreg(ptr, 2), ' offset:%d' % o if o else ''),
                         'V%d' % xset if self.bf else 'V')
        for p in (S_L0P, S_TSP):
            b.e('s_add_u32 %s, %s, 128' % (sreg(p), sreg(p)))
            b.e('s_addc_u32 %s, %s, 0' % (sreg(p + 1), sreg(p + 1)))

    def stage_issue(self, b, slot, reset=False):
        """staged form: LDS-DMA of the next slab of the T / L0 streams into ring slot `slot` - every wave its 1-KiB chunk of the T
        slab, wave 0 the L0 slab.  reset: the streams restart (slab 0 of a column step; the A operand does not depend on the column)"""
        if reset:
            for p, src in ((S_L0P, S_L0), (S_TSP, S_TS)):
                b.e('s_mov_b32 %s, %s' % (sreg(p), sreg(src)))
                b.e('s_mov_b32 %s, %s' % (sreg(p + 1), sreg(src + 1)))
        if 'noreq' not in self.dbg:
            b.e('s_add_u32 m0, %s, %d' % (sreg(S_TCH), TS_OFF + slot * TSLAB))
            b.e('s_nop 0')
            b.vm('global_load_lds_dwordx4 %s, %s' % (vreg(V_TOFF), sreg(S_TSP, 2)), 'T%d' % slot)
            if self.h == 0:
                self.uid += 1
                skip = 'L_r0_l0skip_%d' % self.uid
                b.e('s_cmp_lg_u32 %s, 0' % sreg(S_WAVE))
                b.e('s_cbranch_scc1 %s' % skip)
                b.e('s_mov_b32 m0, %d' % (L0S_OFF + slot * 1024))
                b.e('s_nop 0')
                b.vm_opt('global_load_lds_dwordx4 %s, %s' % (vreg(V_LOFF), sreg(S_L0P, 2)), 'L%d' % slot)
                b.label(skip)
        b.e('s_add_u32 %s, %s, %s' % (sreg(S_TSP), sreg(S_TSP), sreg(S_TSLABB)))
        b.e('s_addc_u32 %s, %s, 0' % (sreg(S_TSP + 1), sreg(S_TSP + 1)))
        b.e('s_add_u32 %s, %s, %d' % (sreg(S_L0P), sreg(S_L0P), 4 * self.slab_k))
        b.e('s_addc_u32 %s, %s, 0' % (sreg(S_L0P + 1), sreg(S_L0P + 1)))

    def stage_read(self, b, slot, xset):
        """this lane's 8 L0 and 8 T values of its k-step of the fragment whose slab sits in `slot` -> register set xset"""
        if 'noreq' in self.dbg:
            return
        if not self.bf:
            xset = 0                                       # hs: one register set, the producer of a fragment reads it
        b.e('ds_read_b128 %s, %s offset:%d' % (vreg(self.tv + 8 * xset, 4), vreg(V_TL0), slot * TSLAB))
        b.e('ds_read_b128 %s, %s offset:%d' % (vreg(self.tv + 8 * xset + 4, 4), vreg(V_TL1), slot * TSLAB))
        b.e('ds_read_b128 %s, %s offset:%d' % (vreg(V_LV + 8 * xset, 4), vreg(V_LL), slot * 1024))
        b.e('ds_read_b128 %s, %s offset:%d' % (vreg(V_LV + 8 * xset + 4, 4), vreg(V_LL), slot * 1024 + 16))

    def pieces(self, b, kind, slot, who='own'):
        """LDS-DMA pieces of the next sub-tile of the stream (kind 's1' / 's2') into ring slot `slot`: who = 'own' (this
        wave's 2), 'all' (its own and its partner's: the wave that does not convert in this sub-step takes both), 'none'"""
        ptr, vo = (S_W1P, V_VO1) if kind == 's1' else (S_W2P, V_VO2)
        if 'nodma' not in self.dbg and who != 'none':
            for p in range(4 if who == 'all' else 2):
                b.e('s_add_u32 m0, %s, %d' % (sreg(S_DMA if p < 2 else S_DMA2), slot * RING_SLOT + (p & 1) * 1024))
                b.e('s_nop 0')
                b.vm('global_load_lds_dwordx4 %s, %s' % (vreg(vo + p), sreg(ptr, 2)), 'P%d' % slot)
                b.items.append(('unit',))            # (p2first: what may be placed between two MFMAs as one piece)
        b.e('s_add_u32 %s, %s, 64' % (sreg(ptr), sreg(ptr)))
        b.e('s_addc_u32 %s, %s, 0' % (sreg(ptr + 1), sreg(ptr + 1)))
        b.items.append(('unit',))

    def convert(self, b, kind, par_next, jj=0, g=0):
        """fragment of the NEXT sub-step into the two operand registers quads [par_next] and the exchange slot.
        hs: (hi, lo) halves of 16 k.  kind 't1': from the requested L0 / Ts values; 't2': from stage-1 accumulator tile jj,
        registers 8 g .. 8 g + 7.  bf16: (k-step 0, k-step 1) of 32 k; 't2' converts the whole tile jj (both halves)."""
        ahi, alo = V_AHI[par_next], V_ALO[par_next]
        pk = V_PK1 if kind == 't1' else V_PK2
        if self.bf and kind == 't1':
            # split production: this wave converts ITS k-step of the fragment (half 0: k-step 0 -> a0, half 1: k-step 1 -> a1),
            # leaves it in the exchange slot and reads the partner's half behind the barrier
            dst = ahi if self.h == 0 else alo
            if 'noconv' not in self.dbg:
                if not self.staged:
                    b.wait_vm({'V%d' % g})             # g = register set of this fragment
                # packed: 4 adds on the fp32 pairs, 4 conversions, relu on the PACKED bf16 as a signed 16-bit maximum with 0 (a negative
                # bf16 is a negative int16, -0 included; rounding commutes with relu) - 12 operations instead of 20, same bits
                for p in range(4):
                    b.e('v_pk_add_f32 %s, %s, %s' % (vreg(V_GV + 2 * p, 2), vreg(V_LV + 8 * g + 2 * p, 2), vreg(self.tv + 8 * g + 2 * p, 2)))
                for p in range(4):
                    b.e('v_cvt_pk_bf16_f32 %s, %s, %s' % (vreg(dst + p), vreg(V_GV + 2 * p), vreg(V_GV + 2 * p + 1)))
                for p in range(4):
                    b.e('v_pk_max_i16 %s, %s, 0' % (vreg(dst + p), vreg(dst + p)))
            b.e('ds_write_b128 %s, %s offset:%d' % (vreg(V_AX), vreg(dst, 4), par_next * 2048 + 1024 * self.h))
            return
        if 'noconv' not in self.dbg and self.bf:
            for half, dst in ((0, ahi), (1, alo)):
                if True:
                    imm = (128 * self.h + 32 * jj + 16 * half) * 4
                    b.e('ds_read_b128 %s, %s offset:%d' % (vreg(V_BQ, 4), vreg(V_BADDR), imm))
                    b.e('ds_read_b128 %s, %s offset:%d' % (vreg(V_BQ + 4, 4), vreg(V_BADDR), imm + 32))
                    for e in range(8):
                        b.e('v_accvgpr_read_b32 %s, %s' % (vreg(V_GV + e), areg(ACC1 + 16 * jj + 8 * half + e)))
                    b.e('s_waitcnt lgkmcnt(0)')
                    for p in range(4):
                        b.e('v_pk_add_f32 %s, %s, %s' % (vreg(V_GV + 2 * p, 2), vreg(V_GV + 2 * p, 2), vreg(V_BQ + 2 * p, 2)))
                for p in range(4):
                    b.e('v_cvt_pk_bf16_f32 %s, %s, %s' % (vreg(dst + p), vreg(V_GV + 2 * p), vreg(V_GV + 2 * p + 1)))
                for p in range(4):
                    b.e('v_pk_max_i16 %s, %s, 0' % (vreg(dst + p), vreg(dst + p)))
        elif 'noconv' not in self.dbg:
            if kind == 't1':
                if not self.staged:
                    b.wait_vm({'V'})
                for e in range(8):
                    b.e('v_fma_f32 %s, %s, %s, %s' % (vreg(V_GV + e), vreg(V_LV + e), sreg(S_INSC), vreg(V_TV + e)))
            else:
                imm = (128 * self.h + 32 * jj + 16 * g) * 4
                b.e('ds_read_b128 %s, %s offset:%d' % (vreg(V_BQ, 4), vreg(V_BADDR), imm))
                b.e('ds_read_b128 %s, %s offset:%d' % (vreg(V_BQ + 4, 4), vreg(V_BADDR), imm + 32))
                for e in range(8):
                    b.e('v_accvgpr_read_b32 %s, %s' % (vreg(V_GV + e), areg(ACC1 + 16 * jj + 8 * g + e)))
                b.e('s_waitcnt lgkmcnt(0)')
                for e in range(8):
                    b.e('v_fma_f32 %s, %s, %s, %s' % (vreg(V_GV + e), vreg(V_GV + e), sreg(S_AS1OS), vreg(V_BQ + e)))
            for e in range(8):
                b.e('v_max_f32_e32 %s, 0, %s' % (vreg(V_GV + e), vreg(V_GV + e)))
            for p in range(4):
                b.e('v_cvt_pk_f16_f32 %s, %s, %s' % (vreg(ahi + p), vreg(V_GV + 2 * p), vreg(V_GV + 2 * p + 1)))
            # lo halves: f16(x - hi) through v_fma_mix (hs_lo_pair); dependent op_sel / packed operations are never adjacent
            for p in range(4):
                b.e('v_fma_mixlo_f16 %s, %s, -1.0, %s op_sel_hi:[1,0,0]' % (vreg(alo + p), vreg(ahi + p), vreg(V_GV + 2 * p)))
                b.e('v_pk_max_u16 %s, %s, %s' % (vreg(pk), vreg(pk), vreg(ahi + p)))
            for p in range(4):
                b.e('v_fma_mixhi_f16 %s, %s, -1.0, %s op_sel:[1,0,0] op_sel_hi:[1,0,0]' % (vreg(alo + p), vreg(ahi + p), vreg(V_GV + 2 * p + 1)))
        b.e('ds_write_b128 %s, %s offset:%d' % (vreg(V_AX), vreg(ahi, 4), par_next * 2048))
        b.e('ds_write_b128 %s, %s offset:%d' % (vreg(V_AX), vreg(alo, 4), par_next * 2048 + 1024))

    def read_frag(self, b, par_next, partner_half_only=False):
        if not partner_half_only or self.h == 1:
            b.e('ds_read_b128 %s, %s offset:%d' % (vreg(V_AHI[par_next], 4), vreg(V_AX), par_next * 2048))
        if not partner_half_only or self.h == 0:
            b.e('ds_read_b128 %s, %s offset:%d' % (vreg(V_ALO[par_next], 4), vreg(V_AX), par_next * 2048 + 1024))

    def fill_operands(self, b):
        """'rnd' (with the skeleton flags: no conversion, requests, LDS-DMA or fragment reads): the 48 operand registers the MFMAs of
        the main loop read are filled ONCE with real operand data, so that the MFMA + barrier skeleton runs at the power point of the
        real kernel instead of on whatever the registers held (round-3 verdict, Missing 6).  Weight fragments: the hi | lo halves of
        W1 itself (row 4 w + jj, 16-k group `lane`: bytes 0-15 hi, 32-47 lo).  Activation fragments: the next hi halves with the
        negative ones cleared (the exact zeros of a relu, about half of them) and the lo halves cleared where the hi half is."""
        b.e('v_lshlrev_b32_e32 %s, 6, %s' % (vreg(V_T), vreg(V_LANE)))
        b.e('s_lshl_b32 %s, %s, 2' % (sreg(S_T), sreg(S_WAVE)))
        b.e('s_mul_i32 %s, %s, %s' % (sreg(S_T), sreg(S_T), sreg(S_LDB1)))            # row 4 w: ldb1 halves per row
        b.e('s_lshl_b32 %s, %s, 1' % (sreg(S_T), sreg(S_T)))
        b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_T), sreg(S_T), vreg(V_T)))
        b.e('s_lshl_b32 %s, %s, 1' % (sreg(S_T + 1), sreg(S_LDB1)))                  # bytes per row
        for jj in range(4):
            b.e('global_load_dwordx4 %s, %s, %s' % (vreg(V_WHI + 4 * jj, 4), vreg(V_T), sreg(S_W1, 2)))
            b.e('global_load_dwordx4 %s, %s, %s offset:32' % (vreg(V_WLO + 4 * jj, 4), vreg(V_T), sreg(S_W1, 2)))
            if jj < 2:
                b.e('global_load_dwordx4 %s, %s, %s offset:16' % (vreg(V_AHI[jj], 4), vreg(V_T), sreg(S_W1, 2)))
                b.e('global_load_dwordx4 %s, %s, %s offset:48' % (vreg(V_ALO[jj], 4), vreg(V_T), sreg(S_W1, 2)))
            b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_T), sreg(S_T + 1), vreg(V_T)))
        b.e('s_waitcnt vmcnt(0)')
        b.e('v_mov_b32_e32 %s, 0x3c003c00' % vreg(V_T + 1))
        for par in range(2):
            for p in range(4):
                b.e('v_pk_max_i16 %s, %s, 0' % (vreg(V_AHI[par] + p), vreg(V_AHI[par] + p)))
                if self.bf:                                  # bf16: both register quads are k-steps of one relu output
                    b.e('v_pk_max_i16 %s, %s, 0' % (vreg(V_ALO[par] + p), vreg(V_ALO[par] + p)))
                    continue
                b.e('v_pk_min_u16 %s, %s, %s' % (vreg(V_T + 2), vreg(V_AHI[par] + p), vreg(V_T + 1)))
                b.e('v_pk_mul_f16 %s, %s, %s' % (vreg(V_ALO[par] + p), vreg(V_ALO[par] + p), vreg(V_T + 2)))

    def barrier(self, b):
        if 'nobarrier' not in self.dbg:
            b.e('s_barrier')

    def substep(self, b, kind, slot, par, first=False, produce=None, consume=False, request=None, piece='s1', pre_piece=None, who='own', stage=None):
        """kind 1 / 2: which accumulator set this sub-step's 12 MFMAs feed"""
        acc = ACC1 if kind == 1 else ACC2
        nslot = (slot + 1) & 3
        b.e('s_waitcnt lgkmcnt(0)')                       # w_lo of this sub-tile, the fragment read behind the last barrier
        self.read_w(b, 0, slot)                           # w_hi: the sub-tile is visible since the last barrier
        # hs: the producer's conversion is dripped between its own MFMAs of P0 (and P1 for half 0) instead of standing in front of
        # them - measured 1.604 against 1.631 ms per 262144 rows, band stamps 355 k against 373 k cycles ('nointerleave' = the block form)
        pending = []
        if produce is not None:
            if 'nointerleave' not in self.dbg and not self.bf:
                tmp = Block('conv')
                self.convert(tmp, produce[0], par ^ 1, *produce[1:])
                pending = list(tmp.items)
            else:
                self.convert(b, produce[0], par ^ 1, *produce[1:])
        if request is not None and not pending:
            if isinstance(request, tuple):
                self.request(b, request[0], request[1])
            else:
                self.request(b, request)
        # hs: P0 w_lo x a_hi, P1 w_hi x a_hi, P2 w_hi x a_lo (three products of the split operands).  bf16: the sub-tile is 32 k =
        # two MFMA k-steps: P0 = k-step 1 (weight chunks 2, 3 = "plane 1", read behind the previous barrier), P1 = k-step 0
        def drip(n):
            for _ in range(n):
                if pending:
                    b.items.append(pending.pop(0))

        def P(ph):
            for jj in range(4):
                drip((len(pending) + 3) // 4 if (self.h == 1 and ph == 0 and 'stagger' in self.dbg) else (len(pending) + (7 - 4 * ph - jj)) // max(8 - 4 * ph - jj, 1) if ph < 2 else 0)
                if self.bf:
                    if ph == 0:
                        self.mfma(b, acc + 16 * jj, V_WLO + 4 * jj, V_ALO[par], zero_c=first)
                    else:
                        self.mfma(b, acc + 16 * jj, V_WHI + 4 * jj, V_AHI[par])
                elif ph == 0:
                    self.mfma(b, acc + 16 * jj, V_WLO + 4 * jj, V_AHI[par], zero_c=first)
                elif ph == 1:
                    self.mfma(b, acc + 16 * jj, V_WHI + 4 * jj, V_AHI[par])
                else:
                    self.mfma(b, acc + 16 * jj, V_WHI + 4 * jj, V_ALO[par])

        def sync():
            drip(len(pending))
            if 'nointerleave' not in self.dbg and request is not None and produce is not None and not self.bf:
                self.request(b, request)
            b.e('s_waitcnt lgkmcnt(0)')                   # w_hi in registers (the slot may be refilled), exchange slot written
            needs = {'P%d' % nslot}
            if stage and stage.get('read') is not None:   # ... and its chunk of the T / L0 slab that is read behind this barrier
                needs |= {'T%d' % stage['read'][0], 'L%d' % stage['read'][0]}
            b.wait_vm(needs)                              # this wave's pieces of the next sub-tile have landed
            self.barrier(b)                               # -> sub-tile s + 1 and fragment s + 1 visible, slot s free

        def after():
            if pre_piece:
                for ln in pre_piece:
                    b.e(ln)
            self.pieces(b, piece, slot, who)
            self.read_w(b, 1, nslot)
            if consume:
                self.read_frag(b, par ^ 1, partner_half_only=(consume == 'half'))
            if stage and stage.get('read') is not None and (len(stage['read']) < 3 or stage['read'][2] == self.h):
                self.stage_read(b, stage['read'][0], stage['read'][1])
            if stage and stage.get('issue') is not None:
                self.stage_issue(b, stage['issue'][0], reset=stage['issue'][1])

        # The two waves of a SIMD (half 0 / half 1 of a row group) meet at ONE barrier per sub-step but sit at different
        # places of their MFMA sequence when they do: half 0 has 8 of its 12 MFMAs in front of it, half 1 four - so one
        # wave's reads / LDS-DMA / conversion run beside the other's MFMAs instead of beside its reads.
        if self.bf and (self.h == 0 or 'nostagger' in self.dbg):
            P(0)
            sync()
            after()
            P(1)
        elif self.bf:
            # half 1 meets the barrier BEFORE its first MFMA group: behind the barrier it runs P0 while half 0 issues its LDS-DMA and
            # reads, then its own LDS-DMA / reads while half 0 runs P1, then P1 while half 0 converts - with 8 MFMAs per wave and
            # barrier there is nothing else to hide a wave's non-MFMA work behind
            sync()
            P(0)
            after()
            P(1)
        elif 'p2first' in self.dbg:
            # Round 4.  Behind the barrier BOTH waves of a SIMD used to issue their LDS-DMA pieces, weight / fragment / slab reads
            # and pointer arithmetic (after()) before the first MFMA of P2 - a few hundred cycles per sub-step in which the SIMD's
            # matrix pipe had nothing queued by either wave.  P2's operands (w_hi, a_lo of this sub-step) sit in registers since
            # before the barrier: its first MFMA goes first, and the post-barrier work is dealt out behind its MFMAs - the LDS reads the
            # next sub-step's P0 waits for first, the LDS-DMA issue last.
            P(0)
            b.e('s_waitcnt lgkmcnt(0)')
            P(1)
            sync()
            tmp = Block('after')
            self.read_w(tmp, 1, nslot)                    # (1) what the next sub-step's P0 waits for
            if consume:
                self.read_frag(tmp, par ^ 1, partner_half_only=(consume == 'half'))
            tmp.items.append(('unit',))
            if stage and stage.get('read') is not None and (len(stage['read']) < 3 or stage['read'][2] == self.h):
                self.stage_read(tmp, stage['read'][0], stage['read'][1])
                tmp.items.append(('unit',))
            if pre_piece:
                for ln in pre_piece:
                    tmp.e(ln)
            self.pieces(tmp, piece, slot, who)            # (2) the LDS-DMA issue: one unit per piece
            if stage and stage.get('issue') is not None:
                self.stage_issue(tmp, stage['issue'][0], reset=stage['issue'][1])
                tmp.items.append(('unit',))
            units, cur = [], []
            for it in tmp.items:
                if it == ('unit',):
                    if cur:
                        units.append(cur)
                    cur = []
                else:
                    cur.append(it)
            if cur:
                units.append(cur)
            for jj in range(4):
                self.mfma(b, acc + 16 * jj, V_WHI + 4 * jj, V_ALO[par])
                take = (len(units) + (3 - jj)) // (4 - jj)
                for _ in range(take):
                    b.items.extend(units.pop(0))
        elif self.h == 0 or 'stagger' not in self.dbg:
            # split-f16 form: both halves in the same order.  (The staggered order of the bf16 form - 'stagger' - was the default until the
            # conversion was dripped between the MFMAs and the operand streams staged: since then 1.650 against 1.662 ms per 262144 rows)
            P(0)
            b.e('s_waitcnt lgkmcnt(0)')
            P(1)
            sync()
            after()
            P(2)
        else:
            P(0)
            sync()
            P(1)
            after()
            P(2)

    # ---------------------------------------------------------------- the role's program
    def build(self):
        h = self.h
        L = lambda s: 'L_r%d_%s' % (h, s)
        pro, head, loop, last, bub, st2, tail = (Block(n) for n in ('pro', 'head', 'loop', 'last', 'bub', 'st2', 'tail'))

        # ---- prologue: the state a column step starts from
        b = pro
        b.label(L('start'))
        if ('prio1' in self.dbg and h == 1) or ('prio0' in self.dbg and h == 0):
            b.e('s_setprio 1')                              # static priority for one half of the workgroup (A/B: MI355X_MICROARCH.md, two waves per SIMD, item 4)
        for p, src in ((S_W1P, S_W1), (S_W2P, S_W2)):
            b.e('s_mov_b32 %s, %s' % (sreg(p), sreg(src)))
            b.e('s_mov_b32 %s, %s' % (sreg(p + 1), sreg(src + 1)))
        if 'rnd' in self.dbg:
            self.fill_operands(b)
        for t in range(4):
            self.pieces(b, 's1', t)
        if self.staged:                                     # slabs 0 .. 3 of the T / L0 streams; fragment 0 from slab 0
            for t in range(4):
                self.stage_issue(b, t, reset=(t == 0))
            b.wait_vm({'T0', 'L0'})
            b.e('s_waitcnt lgkmcnt(0)')
            self.barrier(b)
            if self.bf or h == 0:                           # (hs: half 0 produces fragment 0)
                self.stage_read(b, 0, 0)
                b.e('s_waitcnt lgkmcnt(0)')
                if self.bf:
                    self.convert(b, 't1', 0, 0, 0)
                else:
                    self.convert(b, 't1', 0)
        elif self.bf:                                       # split production: both halves convert their k-step of fragment 0
            self.request(b, 'reset', 0)
            self.request(b, 'advance', 1)
            self.convert(b, 't1', 0, 0, 0)
            self.request(b, 'advance', 0)
        elif h == 0:
            self.request(b, 'reset')
            self.convert(b, 't1', 0)
            self.request(b, 'advance')
        else:
            self.request(b, 'reset')
        b.wait_vm({'P0', 'T1', 'L1'} if self.staged else {'P0'})
        b.e('s_waitcnt lgkmcnt(0)')
        self.barrier(b)
        if self.bf:
            self.read_frag(b, 0, partner_half_only=True)
        elif h == 1:
            self.read_frag(b, 0)
        self.read_w(b, 1, 0)
        if self.staged:                                     # fragment 1 <- slab 1; slab 0 has been read by every wave: slab 4 takes its slot
            if self.bf or h == 1:                           # (hs: half 1 produces fragment 1)
                self.stage_read(b, 1, 1)
            self.stage_issue(b, 0)
        b.e('s_mov_b32 %s, 0' % sreg(S_COL))

        # ---- stage 1.  Fragment u + 1 is produced during sub-step u by half (u + 1) & 1; its producer then requests the
        # values of fragment u + 3.  u_rel = position inside a trip of four, flags say what still exists near the column end.
        def s1_substep(b, i, first=False, produce_ok=True, request_ok=True, piece='s1', stage_ok=True, issue_ok=True):
            if self.bf:
                # fragment u + 1 (register set (i + 1) & 1): both halves convert their k-step; then the values of fragment u + 3
                # staged form: behind the barrier of sub-step u slab u + 2 is read (fragment u + 2, set u & 1) and slab u + 5 requested
                stage = None
                if self.staged:
                    stage = {'read': ((i + 2) & 3, i & 1) if stage_ok else None, 'issue': ((i + 1) & 3, False) if issue_ok else None}
                self.substep(b, 1, i & 3, i & 1, first=first, produce=('t1', 0, (i + 1) & 1) if produce_ok else None,
                             consume='half' if produce_ok else False, request=('advance', (i + 1) & 1) if (produce_ok and request_ok) else None,
                             piece=piece, who='own', stage=stage)
                return
            prod = ((i + 1) & 1) == h and produce_ok
            cons = ((i + 1) & 1) != h and produce_ok
            who = 'none' if prod else ('all' if cons else 'own')       # the converting wave leaves the LDS-DMA to its partner
            if 'ownpieces' in self.dbg:
                who = 'own'
            stage = None
            if self.staged:     # behind the barrier of sub-step u the producer of fragment u + 2 (half u & 1) reads slab u + 2; slab u + 5 is requested
                stage = {'read': ((i + 2) & 3, 0, i & 1) if stage_ok else None, 'issue': ((i + 1) & 3, False) if issue_ok else None}
            self.substep(b, 1, i & 3, i & 1, first=first, produce=('t1',) if prod else None, consume=cons,
                         request='advance' if (prod and request_ok) else None, piece=piece, who=who, stage=stage)

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
            s1_substep(last, i, produce_ok=(i < 3), request_ok=(i == 0), piece='s2', stage_ok=(i < 2), issue_ok=False)

        # ---- first fragment of stage 2 (features 0..15 of the column step, tile 0 of half 0) needs the finished accumulators
        if h == 0:
            bub.e('s_nop 15')
            bub.e('s_nop 15')
            self.convert(bub, 't2', 0, 0, 0)                # (bf16: the whole tile 0)
            bub.e('s_waitcnt lgkmcnt(0)')
            self.barrier(bub)
        else:
            self.barrier(bub)
            self.read_frag(bub, 0)

        # ---- stage 2: sub-step q consumes fragment q (hs: tile q >> 1, half q & 1 of its 16-feature groups; bf16: tile q)
        NQ = self.nq
        for q in range(NQ):
            nq = q + 1
            produce, consume, request, pre = None, False, None, None
            if nq <= NQ - 1:
                owner = 0 if nq < NQ // 2 else 1
                if owner == h:
                    produce = ('t2', nq - 4 * h) if self.bf else ('t2', (nq >> 1) - 4 * h, nq & 1)
                else:
                    consume = True
            elif self.bf:                                  # fragment 0 of the next column step: split production (register set 0)
                produce, consume, request = ('t1', 0, 0), 'half', ('advance', 0)
            else:                                          # fragment 0 of the next column step (stage-1 kind), by half 0
                if h == 0:
                    produce, request = ('t1',), 'advance'
                else:
                    consume = True
            if self.bf:
                if q == NQ - 3:
                    request = ('reset', 0)
                if q == NQ - 2:
                    request = ('advance', 1)
            else:
                if q == NQ - 3 and h == 0:
                    request = 'reset'
                if q == NQ - 2 and h == 1:
                    request = 'reset'
            piece = 's2' if q < NQ - 4 else 's1'
            if q == NQ - 4:                                # weight pointer of stage 1 moves to the next column step (wraps at the end)
                pre = ['s_add_u32 %s, %s, 1' % (sreg(S_T), sreg(S_COL)),
                       's_cmp_ge_u32 %s, %s' % (sreg(S_T), sreg(S_NCOL)),
                       's_cselect_b32 %s, 0, %s' % (sreg(S_T), sreg(S_T)),
                       's_mul_i32 %s, %s, %s' % (sreg(S_T + 1), sreg(S_T), sreg(S_COLBYTES)),
                       's_add_u32 %s, %s, %s' % (sreg(S_W1P), sreg(S_W1), sreg(S_T + 1)),
                       's_addc_u32 %s, %s, 0' % (sreg(S_W1P + 1), sreg(S_W1 + 1))]
            who = 'none' if produce is not None else ('all' if consume else 'own')
            if 'ownpieces' in self.dbg or consume == 'half':      # split production: both halves convert, both issue their own pieces
                who = 'own'
            stage = None
            if self.staged:                                # the next column step's slabs 0 .. 4, fragments 0 and 1
                r0, r1 = ((0, 0), (1, 1)) if self.bf else ((0, 0, 0), (1, 0, 1))      # hs: fragment 0 by half 0, fragment 1 by half 1
                stage = {NQ - 5: {'issue': (0, True)}, NQ - 4: {'issue': (1, False)}, NQ - 3: {'issue': (2, False)},
                         NQ - 2: {'read': r0, 'issue': (3, False)}, NQ - 1: {'read': r1, 'issue': (0, False)}}.get(q)
            self.substep(st2, 2, q & 3, q & 1, produce=produce, consume=consume, request=request, piece=piece, pre_piece=pre, who=who, stage=stage)
        tail.e('v_add_u32_e32 %s, 1024, %s' % (vreg(V_BADDR), vreg(V_BADDR)))
        tail.e('s_add_u32 %s, %s, 1' % (sreg(S_COL), sreg(S_COL)))
        tail.e('s_cmp_lt_u32 %s, %s' % (sreg(S_COL), sreg(S_NCOL)))
        tail.e('s_cbranch_scc1 %s' % L('col'))
        tail.e('s_branch L_epilogue_%d' % h)

        # ---- counted waits: every path through the loops
        col0 = [head, last, bub, st2, tail]
        col1 = [head, loop, last, bub, st2, tail]
        col2 = [head, loop, loop, last, bub, st2, tail]
        for first_col in (col0, col1, col2):
            for second_col in (col0, col1, col2):
                for with_opt in (True, False):
                    simulate([pro] + first_col + second_col + second_col, with_opt=with_opt)
        return [pro, head, loop, last, bub, st2, tail]


def stamp(b, i, uid):
    """wave 0, lane 0: (shader cycles, wall ticks) into stamps[(workgroup * 6 + i) * 2 ..]"""
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


def common_prologue(b, dbg=(), mode='hs'):
    b.e('s_load_dwordx16 %s, s[0:1], 0x0' % sreg(4, 16))
    b.e('s_load_dwordx16 %s, s[0:1], 0x40' % sreg(20, 16))
    b.e('s_waitcnt lgkmcnt(0)')
    b.e('v_and_b32_e32 %s, 63, v0' % vreg(V_LANE))
    b.e('v_readfirstlane_b32 %s, v0' % sreg(S_WAVE))
    b.e('s_nop 4')
    b.e('s_lshr_b32 %s, %s, 6' % (sreg(S_WAVE), sreg(S_WAVE)))
    b.e('s_and_b32 %s, %s, 3' % (sreg(S_RG), sreg(S_WAVE)))
    b.e('s_lshr_b32 %s, %s, 2' % (sreg(S_H), sreg(S_WAVE)))
    b.e('s_lshl_b32 %s, s2, 7' % sreg(S_M0))
    if 'dump0' in dbg:         # bring-up: count the waves that arrive, leave a few raw registers
        b.e('s_mov_b64 exec, 1')
        b.e('v_mov_b32_e32 %s, 0' % vreg(V_T))
        b.e('v_mov_b32_e32 %s, 1' % vreg(V_T + 1))
        b.e('global_atomic_add %s, %s, %s' % (vreg(V_T), vreg(V_T + 1), sreg(S_PEAK, 2)))
        for k, src in enumerate(('s2', 's3', sreg(S_M), sreg(S_WAVE), sreg(S_RG), sreg(S_H), sreg(S_M0), sreg(S_MAGIC))):
            b.e('v_mov_b32_e32 %s, %s' % (vreg(V_T + 1), src))
            b.e('global_store_dword %s, %s, %s offset:%d' % (vreg(V_T), vreg(V_T + 1), sreg(S_PEAK, 2), 4 + 4 * k))
        b.e('global_store_dword %s, v0, %s offset:%d' % (vreg(V_T), sreg(S_PEAK, 2), 40))
        b.e('s_waitcnt vmcnt(0)')
        b.e('s_endpgm')
    b.e('s_cmp_ge_i32 %s, %s' % (sreg(S_M0), sreg(S_M)))
    b.e('s_cbranch_scc1 L_end')
    if 'colsplit' in dbg:
        # column-split form (small calls: fewer bands than CUs): workgroup (x, y) computes band x over the N1 hidden features
        # [y N1, (y + 1) N1) of a layer that is gridDim.y * N1 wide - the kernel's own arguments describe split 0, split y moves
        # W1 by y N1 rows, bias1 by y N1 entries, the regressor weights by y N1 k-columns (4 bytes each: hi | lo), reads a zero bias2
        # and writes its partial outputs to part + (y - 1) * M * ldo * 4; the host adds the partials to split 0's output in y order
        b.e('s_load_dwordx2 %s, s[0:1], 0x80' % sreg(S_PART, 2))
        b.e('s_waitcnt lgkmcnt(0)')
        b.e('s_cmp_eq_u32 s3, 0')
        b.e('s_cbranch_scc1 L_cs_done')
        b.e('s_lshr_b32 %s, %s, 8' % (sreg(S_T), sreg(S_N1)))
        b.e('s_lshl_b32 %s, %s, 9' % (sreg(S_T + 1), sreg(S_LDB1)))
        b.e('s_mul_i32 %s, %s, %s' % (sreg(S_T), sreg(S_T), sreg(S_T + 1)))            # bytes of the N1 weight rows of one split (host: < 2^31 / splits)
        b.e('s_mul_i32 %s, %s, s3' % (sreg(S_T), sreg(S_T)))
        b.e('s_add_u32 %s, %s, %s' % (sreg(S_W1), sreg(S_W1), sreg(S_T)))
        b.e('s_addc_u32 %s, %s, 0' % (sreg(S_W1 + 1), sreg(S_W1 + 1)))
        b.e('s_mul_i32 %s, %s, s3' % (sreg(S_T), sreg(S_N1)))
        b.e('s_lshl_b32 %s, %s, 2' % (sreg(S_T), sreg(S_T)))                         # bias1: floats; regressor k-columns: hi | lo = 4 bytes (bf16: 2)
        b.e('s_lshr_b32 %s, %s, %d' % (sreg(S_T + 1), sreg(S_T), 1 if mode == 'bf16' else 0))
        for ptr, off in ((S_B1, S_T), (S_W2, S_T + 1)):
            b.e('s_add_u32 %s, %s, %s' % (sreg(ptr), sreg(ptr), sreg(off)))
            b.e('s_addc_u32 %s, %s, 0' % (sreg(ptr + 1), sreg(ptr + 1)))
        b.e('s_mul_i32 %s, %s, %s' % (sreg(S_T), sreg(S_M), sreg(S_LDO)))
        b.e('s_lshl_b32 %s, %s, 2' % (sreg(S_T), sreg(S_T)))
        b.e('s_sub_u32 %s, s3, 1' % sreg(S_T + 1))
        b.e('s_mul_hi_u32 %s, %s, %s' % (sreg(S_T + 2), sreg(S_T), sreg(S_T + 1)))
        b.e('s_mul_i32 %s, %s, %s' % (sreg(S_T), sreg(S_T), sreg(S_T + 1)))
        b.e('s_add_u32 %s, %s, %s' % (sreg(S_OUT), sreg(S_PART), sreg(S_T)))
        b.e('s_addc_u32 %s, %s, %s' % (sreg(S_OUT + 1), sreg(S_PART + 1), sreg(S_T + 2)))
        b.label('L_cs_done')
    if 'exit0' in dbg:
        b.e('s_branch L_end')
    stamp(b, 0, 0)
    b.e('s_lshr_b32 %s, %s, %d' % (sreg(S_NSUB1), sreg(S_K1), 5 if mode == 'bf16' else 4))        # sub-tiles of 16 (hs) / 32 (bf16) k
    b.e('s_lshr_b32 %s, %s, 8' % (sreg(S_NCOL), sreg(S_N1)))
    b.e('s_lshl_b32 %s, %s, 9' % (sreg(S_COLBYTES), sreg(S_LDB1)))        # 256 rows x ldb1 halves x 2 B
    b.e('s_lshl_b32 %s, %s, 11' % (sreg(S_DMA), sreg(S_WAVE)))            # this wave's 2 pieces: image rows 32 w ..
    # ---- bias tables: bias1s[i] = out_scale * bias1[i] (i < N1), bias2s[i] = bias2[i] (i < n2, else 0; 256 entries)
    b.e('s_mov_b32 %s, 0' % sreg(S_T))
    b.label('L_b1')
    b.e('v_add_u32_e32 %s, %s, v0' % (vreg(V_T), sreg(S_T)))
    b.e('v_cmp_gt_u32_e32 vcc, %s, %s' % (sreg(S_N1), vreg(V_T)))
    b.e('s_and_saveexec_b64 %s, vcc' % sreg(S_SAVE, 2))
    b.e('v_lshlrev_b32_e32 %s, 2, %s' % (vreg(V_T + 1), vreg(V_T)))
    b.e('global_load_dword %s, %s, %s' % (vreg(V_T + 2), vreg(V_T + 1), sreg(S_B1, 2)))
    b.e('s_waitcnt vmcnt(0)')
    b.e('v_mul_f32_e32 %s, %s, %s' % (vreg(V_T + 2), sreg(S_OUTSC), vreg(V_T + 2)))
    b.e('v_add_u32_e32 %s, %d, %s' % (vreg(V_T + 1), BIAS1_OFF, vreg(V_T + 1)))
    b.e('ds_write_b32 %s, %s' % (vreg(V_T + 1), vreg(V_T + 2)))
    b.e('s_mov_b64 exec, %s' % sreg(S_SAVE, 2))
    b.e('s_add_u32 %s, %s, 512' % (sreg(S_T), sreg(S_T)))
    b.e('s_cmp_lt_u32 %s, %s' % (sreg(S_T), sreg(S_N1)))
    b.e('s_cbranch_scc1 L_b1')
    b.e('s_lshl_b32 %s, %s, 2' % (sreg(S_BIAS2OFF), sreg(S_N1)))
    b.e('s_add_u32 %s, %s, %d' % (sreg(S_BIAS2OFF), sreg(S_BIAS2OFF), BIAS1_OFF))
    b.e('v_mov_b32_e32 %s, 0' % vreg(V_T + 2))
    b.e('v_lshlrev_b32_e32 %s, 2, v0' % vreg(V_T + 1))
    b.e('v_cmp_gt_u32_e32 vcc, %s, v0' % sreg(S_N2))
    if 'colsplit' in dbg:                                  # splits 1 .. carry no bias2 (the host adds their outputs to split 0's)
        b.e('s_cmp_eq_u32 s3, 0')
        b.e('s_cselect_b64 %s, -1, 0' % sreg(S_T, 2))
        b.e('s_and_b64 vcc, vcc, %s' % sreg(S_T, 2))
    b.e('s_and_saveexec_b64 %s, vcc' % sreg(S_SAVE, 2))
    b.e('global_load_dword %s, %s, %s' % (vreg(V_T + 2), vreg(V_T + 1), sreg(S_B2, 2)))
    b.e('s_waitcnt vmcnt(0)')
    b.e('s_mov_b64 exec, %s' % sreg(S_SAVE, 2))
    b.e('v_cmp_gt_u32_e32 vcc, 256, v0')
    b.e('s_and_saveexec_b64 %s, vcc' % sreg(S_SAVE, 2))
    b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_T + 1), sreg(S_BIAS2OFF), vreg(V_T + 1)))
    b.e('ds_write_b32 %s, %s' % (vreg(V_T + 1), vreg(V_T + 2)))
    b.e('s_mov_b64 exec, %s' % sreg(S_SAVE, 2))
    # ---- this lane's row: m = m0 + 32 rg + l31 (clamped for the loads), pr = m / nt, t = m - pr * nt
    b.e('v_and_b32_e32 %s, 31, %s' % (vreg(V_L31), vreg(V_LANE)))
    b.e('v_lshrrev_b32_e32 %s, 5, %s' % (vreg(V_HI), vreg(V_LANE)))
    b.e('s_lshl_b32 %s, %s, 5' % (sreg(S_T), sreg(S_RG)))
    b.e('s_add_u32 %s, %s, %s' % (sreg(S_T), sreg(S_T), sreg(S_M0)))
    b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_M), sreg(S_T), vreg(V_L31)))
    b.e('v_cmp_gt_i32_e32 vcc, %s, %s' % (sreg(S_M), vreg(V_M)))
    b.e('s_mov_b64 %s, vcc' % sreg(S_ROWMASK, 2))
    b.e('s_sub_u32 %s, %s, 1' % (sreg(S_T), sreg(S_M)))
    b.e('v_min_u32_e32 %s, %s, %s' % (vreg(V_T), sreg(S_T), vreg(V_M)))
    b.e('v_mul_hi_u32 %s, %s, %s' % (vreg(V_T + 1), vreg(V_T), sreg(S_MAGIC)))          # q <= m / nt <= q + 1
    b.e('v_mul_lo_u32 %s, %s, %s' % (vreg(V_T + 2), vreg(V_T + 1), sreg(S_NT)))
    b.e('v_sub_u32_e32 %s, %s, %s' % (vreg(V_T + 3), vreg(V_T), vreg(V_T + 2)))          # r = m - q nt
    b.e('v_cmp_le_u32_e32 vcc, %s, %s' % (sreg(S_NT), vreg(V_T + 3)))
    b.e('v_cndmask_b32_e64 %s, 0, 1, vcc' % vreg(V_T + 4))
    b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_T + 1), vreg(V_T + 1), vreg(V_T + 4)))
    b.e('v_mul_lo_u32 %s, %s, %s' % (vreg(V_T + 4), vreg(V_T + 4), sreg(S_NT)))
    b.e('v_sub_u32_e32 %s, %s, %s' % (vreg(V_T + 3), vreg(V_T + 3), vreg(V_T + 4)))
    if 'nostage' not in dbg:
        # staged form (pr = v[V_T+1], t = v[V_T+3]).  LDS read addresses: T unit u of row t sits at t * 128 + ((u ^ ((t >> 1) & 7)) << 4),
        # this lane's k-step is units 4 h + 2 hi, + 1; the L0 rows of the band sit unswizzled, row pr - pr0
        bf = mode == 'bf16'
        rsh = 7 if bf else 6                      # log2 of the bytes of a slab row (32 / 16 k)
        if bf:
            b.e('v_bfe_u32 %s, %s, 1, 3' % (vreg(V_T + 5), vreg(V_T + 3)))                   # (t >> 1) & 7
            b.e('s_lshl_b32 %s, %s, 2' % (sreg(S_T), sreg(S_H)))
            b.e('v_lshl_add_u32 %s, %s, 1, %s' % (vreg(V_T + 6), vreg(V_HI), sreg(S_T)))     # unit 4 h + 2 hi: this half's k-step
        else:                                     # hs: slab rows of 16 k = 4 units, swizzle (t >> 2) & 3; the producer converts units 2 hi, 2 hi + 1
            b.e('v_bfe_u32 %s, %s, 2, 2' % (vreg(V_T + 5), vreg(V_T + 3)))
            b.e('v_lshlrev_b32_e32 %s, 1, %s' % (vreg(V_T + 6), vreg(V_HI)))
        b.e('v_lshlrev_b32_e32 %s, %d, %s' % (vreg(V_T + 7), rsh, vreg(V_T + 3)))            # t * row bytes
        b.e('v_add_u32_e32 %s, %d, %s' % (vreg(V_T + 7), TS_OFF, vreg(V_T + 7)))
        b.e('v_xor_b32_e32 %s, %s, %s' % (vreg(V_T + 8), vreg(V_T + 6), vreg(V_T + 5)))
        b.e('v_lshl_add_u32 %s, %s, 4, %s' % (vreg(V_TL0), vreg(V_T + 8), vreg(V_T + 7)))
        b.e('v_or_b32_e32 %s, 1, %s' % (vreg(V_T + 8), vreg(V_T + 6)))
        b.e('v_xor_b32_e32 %s, %s, %s' % (vreg(V_T + 8), vreg(V_T + 8), vreg(V_T + 5)))
        b.e('v_lshl_add_u32 %s, %s, 4, %s' % (vreg(V_TL1), vreg(V_T + 8), vreg(V_T + 7)))
        # pr0 = m0 / nt and prmax = (M - 1) / nt: the same exact division on uniform values
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
        # L0 slab read: row pr - pr0, units 4 h + 2 hi, + 1 (contiguous)
        b.e('v_sub_u32_e32 %s, %s, %s' % (vreg(V_T + 8), vreg(V_T + 1), vreg(V_T + 11)))
        b.e('v_lshlrev_b32_e32 %s, %d, %s' % (vreg(V_T + 8), rsh, vreg(V_T + 8)))
        b.e('v_lshl_add_u32 %s, %s, 4, %s' % (vreg(V_T + 8), vreg(V_T + 6), vreg(V_T + 8)))
        b.e('v_add_u32_e32 %s, %d, %s' % (vreg(V_LL), L0S_OFF, vreg(V_T + 8)))
        # L0 slab DMA (wave 0): lane -> row min(pr0 + (lane >> 3), prmax), 16-byte unit lane & 7
        b.e('v_lshrrev_b32_e32 %s, %d, %s' % (vreg(V_T + 8), 3 if bf else 2, vreg(V_LANE)))
        b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_T + 8), vreg(V_T + 8), vreg(V_T + 11)))
        b.e('v_min_u32_e32 %s, %s, %s' % (vreg(V_T + 8), vreg(V_T + 8), vreg(V_T + 12)))
        b.e('v_mul_lo_u32 %s, %s, %s' % (vreg(V_T + 8), vreg(V_T + 8), sreg(S_LDL)))
        b.e('v_and_b32_e32 %s, %d, %s' % (vreg(V_T + 9), 7 if bf else 3, vreg(V_LANE)))
        b.e('v_lshlrev_b32_e32 %s, 2, %s' % (vreg(V_T + 9), vreg(V_T + 9)))
        b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_T + 8), vreg(V_T + 8), vreg(V_T + 9)))
        b.e('v_lshlrev_b32_e32 %s, 2, %s' % (vreg(V_LOFF), vreg(V_T + 8)))
        # T slab DMA: this wave's 1-KiB chunk min(wave, nch - 1) of the nt * 128 bytes
        b.e('s_lshl_b32 %s, %s, %d' % (sreg(S_TSLABB), sreg(S_NT), rsh))
        b.e('s_add_u32 %s, %s, %d' % (sreg(S_T), sreg(S_NT), 7 if bf else 15))
        b.e('s_lshr_b32 %s, %s, %d' % (sreg(S_T), sreg(S_T), 3 if bf else 4))
        b.e('s_sub_u32 %s, %s, 1' % (sreg(S_T), sreg(S_T)))
        b.e('s_min_u32 %s, %s, %s' % (sreg(S_T), sreg(S_T), sreg(S_WAVE)))
        b.e('s_lshl_b32 %s, %s, 10' % (sreg(S_TCH), sreg(S_T)))
        b.e('v_lshlrev_b32_e32 %s, 4, %s' % (vreg(V_TOFF), vreg(V_LANE)))
        b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_TOFF), sreg(S_TCH), vreg(V_TOFF)))
    else:
        for dst, idx in ((V_LOFF, V_T + 1), (V_TOFF, V_T + 3)):
            b.e('v_mul_lo_u32 %s, %s, %s' % (vreg(V_T + 5), vreg(idx), sreg(S_LDL)))
            b.e('v_lshl_add_u32 %s, %s, 3, %s' % (vreg(V_T + 5), vreg(V_HI), vreg(V_T + 5)))
            b.e('v_lshlrev_b32_e32 %s, 2, %s' % (vreg(dst), vreg(V_T + 5)))
    # ---- weight fragment reads: row (128 h + 32 jj + l31) of the image, 16-byte chunk ((2 plane + hi) ^ ((l31 >> 2) & 3))
    b.e('v_bfe_u32 %s, %s, 2, 2' % (vreg(V_T), vreg(V_L31)))
    b.e('v_xor_b32_e32 %s, %s, %s' % (vreg(V_T + 1), vreg(V_HI), vreg(V_T)))
    b.e('v_or_b32_e32 %s, 2, %s' % (vreg(V_T + 2), vreg(V_HI)))
    b.e('v_xor_b32_e32 %s, %s, %s' % (vreg(V_T + 2), vreg(V_T + 2), vreg(V_T)))
    b.e('v_lshlrev_b32_e32 %s, 6, %s' % (vreg(V_T + 3), vreg(V_L31)))
    b.e('s_lshl_b32 %s, %s, 13' % (sreg(S_T), sreg(S_H)))
    b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_T + 3), sreg(S_T), vreg(V_T + 3)))
    b.e('v_lshl_add_u32 %s, %s, 4, %s' % (vreg(V_RDHI), vreg(V_T + 1), vreg(V_T + 3)))
    b.e('v_lshl_add_u32 %s, %s, 4, %s' % (vreg(V_RDLO), vreg(V_T + 2), vreg(V_T + 3)))
    # ---- fragment exchange slot of this row group: lane-linear 16 B
    b.e('v_lshlrev_b32_e32 %s, 4, %s' % (vreg(V_AX), vreg(V_LANE)))
    b.e('s_lshl_b32 %s, %s, 12' % (sreg(S_T), sreg(S_RG)))
    b.e('s_add_u32 %s, %s, %d' % (sreg(S_T), sreg(S_T), AX_OFF))
    b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_AX), sreg(S_T), vreg(V_AX)))
    # ---- LDS-DMA pieces: image row 32 w + 16 p + (lane >> 2), chunk (lane & 3) ^ ((lane >> 4) & 3); for this wave (w) and
    # for its partner on the SIMD (w ^ 4), whose pieces it issues in the sub-steps in which the partner converts
    b.e('v_and_b32_e32 %s, 3, %s' % (vreg(V_T + 1), vreg(V_LANE)))
    b.e('v_bfe_u32 %s, %s, 4, 2' % (vreg(V_T + 2), vreg(V_LANE)))
    b.e('v_xor_b32_e32 %s, %s, %s' % (vreg(V_T + 1), vreg(V_T + 1), vreg(V_T + 2)))
    b.e('v_lshlrev_b32_e32 %s, 4, %s' % (vreg(V_T + 1), vreg(V_T + 1)))
    b.e('s_xor_b32 %s, %s, 4' % (sreg(S_T + 1), sreg(S_WAVE)))
    b.e('s_lshl_b32 %s, %s, 11' % (sreg(S_DMA2), sreg(S_T + 1)))
    for k, wv in ((0, S_WAVE), (2, S_T + 1)):
        b.e('v_lshrrev_b32_e32 %s, 2, %s' % (vreg(V_T), vreg(V_LANE)))
        b.e('s_lshl_b32 %s, %s, 5' % (sreg(S_T), sreg(wv)))
        b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_T), sreg(S_T), vreg(V_T)))
        for vo, ld in ((V_VO1, S_LDB1), (V_VO2, S_LDB2)):
            b.e('v_mul_lo_u32 %s, %s, %s' % (vreg(V_T + 3), vreg(V_T), sreg(ld)))
            b.e('v_lshl_add_u32 %s, %s, 1, %s' % (vreg(vo + k), vreg(V_T + 3), vreg(V_T + 1)))
            b.e('s_lshl_b32 %s, %s, 5' % (sreg(S_T), sreg(ld)))
            b.e('v_add_u32_e32 %s, %s, %s' % (vreg(vo + k + 1), sreg(S_T), vreg(vo + k)))
    # ---- bias reads, output addressing, guards
    b.e('v_lshlrev_b32_e32 %s, 4, %s' % (vreg(V_BADDR), vreg(V_HI)))
    b.e('v_add_u32_e32 %s, %d, %s' % (vreg(V_BADDR), BIAS1_OFF, vreg(V_BADDR)))
    b.e('v_lshlrev_b32_e32 %s, 2, %s' % (vreg(V_4HI), vreg(V_HI)))
    b.e('s_lshl_b32 %s, %s, 9' % (sreg(S_T), sreg(S_H)))                                 # 128 h floats = 512 h bytes
    b.e('v_lshl_add_u32 %s, %s, 4, %s' % (vreg(V_B2ADDR), vreg(V_HI), sreg(S_BIAS2OFF)))
    b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_B2ADDR), sreg(S_T), vreg(V_B2ADDR)))
    b.e('v_mul_lo_u32 %s, %s, %s' % (vreg(V_OUTOFF), vreg(V_M), sreg(S_LDO)))
    b.e('v_lshlrev_b32_e32 %s, 2, %s' % (vreg(V_OUTOFF), vreg(V_OUTOFF)))
    b.e('v_lshl_add_u32 %s, %s, 4, %s' % (vreg(V_OUTOFF), vreg(V_HI), vreg(V_OUTOFF)))
    b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_OUTOFF), sreg(S_T), vreg(V_OUTOFF)))
    b.e('v_mov_b32_e32 %s, 0' % vreg(V_PK1))
    b.e('v_mov_b32_e32 %s, 0' % vreg(V_PK2))
    for i in range(64):
        b.e('v_accvgpr_write_b32 %s, 0' % areg(ACC2 + i))
    b.e('s_waitcnt lgkmcnt(0)')
    stamp(b, 1, 1)
    if 'exit1' in dbg:
        b.e('s_branch L_end')
    if 'dump' in dbg:          # bring-up: workgroup 0, wave 0, lane 0 writes a few registers to the range-guard words
        b.e('s_cmp_lg_u32 s2, 0')
        b.e('s_cbranch_scc1 L_end')
        b.e('s_cmp_lg_u32 %s, 0' % sreg(S_WAVE))
        b.e('s_cbranch_scc1 L_end')
        b.e('s_mov_b64 exec, 1')
        b.e('v_mov_b32_e32 %s, 0' % vreg(V_T))
        for k, src in enumerate((sreg(S_OUT), sreg(S_OUT + 1), sreg(S_LDO), sreg(S_M), sreg(S_N2), sreg(S_PEAK), sreg(S_PEAK + 1), sreg(S_STAMPS))):
            b.e('v_mov_b32_e32 %s, %s' % (vreg(V_T + 1), src))
            b.e('global_store_dword %s, %s, %s offset:%d' % (vreg(V_T), vreg(V_T + 1), sreg(S_PEAK, 2), 4 * k))
        for k, src in enumerate((V_OUTOFF, V_M, V_B2ADDR, V_LOFF, V_TOFF, V_VO1, V_VO2, V_RDHI)):
            b.e('global_store_dword %s, %s, %s offset:%d' % (vreg(V_T), vreg(src), sreg(S_PEAK, 2), 32 + 4 * k))
        b.e('s_waitcnt vmcnt(0)')
        b.e('s_branch L_end')
    if 'exit2' in dbg:
        b.e('s_cmp_eq_u32 %s, 0' % sreg(S_H))
        b.e('s_cbranch_scc1 L_epilogue_0')
        b.e('s_branch L_epilogue_1')
    b.e('s_cmp_eq_u32 %s, 0' % sreg(S_H))
    b.e('s_cbranch_scc0 L_r1_start')


def epilogue_staged(b, h, slow_label):
    """Round 6.  With ldo == n2 (the library's layout) the output of a band is ONE contiguous block of 128 x ldo floats, but a lane owns a
    ROW: the direct stores of the accumulators are 64 scattered 8-byte pieces per instruction - measured on the register-blocked bf16 form
    (band4_kernel_gen.py) at 0.46 ms of a 3.33 ms launch.  Here the block is assembled in LDS (ring, exchange area and slabs are dead; the
    bias quads go to registers first) and leaves as a linear stream of 16-byte pieces, 1 KiB per wave-instruction.  Taken when the band is
    full, ldo == n2, n2 is even (8-byte LDS writes) and the block is 16-byte aligned; anything else branches to the row-per-lane stores."""
    BQ2, ST, D, GOFF, U, NW = 2, V_T + 12, 2, 30, 5, 8
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
    for jj in range(4):                       # bias2 quads of this half's 16 column chunks
        for rq in range(4):
            b.e('ds_read_b128 %s, %s offset:%d' % (vreg(BQ2 + 4 * (4 * jj + rq), 4), vreg(V_B2ADDR), (32 * jj + 8 * rq) * 4))
    # LDS address of this lane's row, its first column: ((32 rg + l31) * ldo + 128 h + 4 hi) * 4
    b.e('s_lshl_b32 %s, %s, 5' % (sreg(S_T), sreg(S_RG)))
    b.e('v_add_u32_e32 %s, %s, %s' % (vreg(ST), sreg(S_T), vreg(V_L31)))
    b.e('v_mul_lo_u32 %s, %s, %s' % (vreg(ST), vreg(ST), sreg(S_LDO)))
    b.e('v_add_u32_e32 %s, %s, %s' % (vreg(ST), vreg(ST), vreg(V_4HI)))
    b.e('v_add_u32_e32 %s, %d, %s' % (vreg(ST), 128 * h, vreg(ST)))
    b.e('v_lshlrev_b32_e32 %s, 2, %s' % (vreg(ST), vreg(ST)))
    b.e('s_waitcnt lgkmcnt(0)')
    b.e('s_barrier')                          # every wave has left the main loop and holds its bias quads: the LDS is the staging area now
    for jj in range(4):
        for rq in range(4):
            c0 = 128 * h + 32 * jj + 8 * rq
            imm = (32 * jj + 8 * rq) * 4
            uid = 'L_s%d_%d_%d' % (h, jj, rq)
            bq = BQ2 + 4 * (4 * jj + rq)
            for e in range(4):
                b.e('v_accvgpr_read_b32 %s, %s' % (vreg(V_T + e), areg(ACC2 + 16 * jj + 4 * rq + e)))
            for e in range(4):
                b.e('v_fma_f32 %s, %s, %s, %s' % (vreg(V_T + e), vreg(V_T + e), sreg(S_AS2), vreg(bq + e)))
            b.e('s_cmp_ge_u32 %s, %d' % (sreg(S_N2), c0 + 8))
            b.e('s_cbranch_scc0 %s_m' % uid)
            b.e('ds_write_b64 %s, %s offset:%d' % (vreg(ST), vreg(V_T, 2), imm))
            b.e('ds_write_b64 %s, %s offset:%d' % (vreg(ST), vreg(V_T + 2, 2), imm + 8))
            b.e('s_branch %s_d' % uid)
            b.label('%s_m' % uid)
            for pr in range(2):                                                         # column pair valid <=> 4 hi + 2 pr < n2 - c0 (n2 even)
                b.e('s_sub_i32 %s, %s, %d' % (sreg(S_T), sreg(S_N2), c0 + 2 * pr))
                b.e('v_cmp_gt_i32_e32 vcc, %s, %s' % (sreg(S_T), vreg(V_4HI)))
                b.e('s_mov_b64 exec, vcc')
                b.e('ds_write_b64 %s, %s offset:%d' % (vreg(ST), vreg(V_T + 2 * pr, 2), imm + 8 * pr))
            b.e('s_mov_b64 exec, -1')
            b.label('%s_d' % uid)
    b.e('s_waitcnt lgkmcnt(0)')
    b.e('s_barrier')
    # ---- the block leaves: piece i (16 bytes) of 32 * ldo; wave w takes pieces 64 w + lane + 512 j
    b.e('s_lshl_b32 %s, %s, 6' % (sreg(S_T), sreg(S_WAVE)))
    b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_T + 8), sreg(S_T), vreg(V_LANE)))          # piece index of round 0
    b.e('v_lshlrev_b32_e32 %s, 4, %s' % (vreg(V_T + 9), vreg(V_T + 8)))                 # its LDS address; advances by U rounds per iteration
    for k in range(U):
        b.e('v_add_u32_e32 %s, %d, %s' % (vreg(GOFF + k), 1024 * NW * k, vreg(V_T + 9)))    # global offsets of the U rounds (the base advances)
    b.e('s_lshl_b32 %s, %s, 5' % (sreg(S_T + 2), sreg(S_LDO)))                          # pieces
    b.e('s_mov_b32 %s, 0' % sreg(S_T + 3))
    b.label('L_copy_%d' % h)
    for k in range(U):
        b.e('v_add_u32_e32 %s, %s, %s' % (vreg(V_T + 10), sreg(S_T + 3), vreg(V_T + 8)))
        if k:
            b.e('v_add_u32_e32 %s, %d, %s' % (vreg(V_T + 10), 64 * NW * k, vreg(V_T + 10)))
        b.e('v_cmp_gt_u32_e32 vcc, %s, %s' % (sreg(S_T + 2), vreg(V_T + 10)))
        b.e('s_mov_b64 %s, vcc' % sreg(78 + 2 * k, 2))
        b.e('s_mov_b64 exec, vcc')
        b.e('ds_read_b128 %s, %s offset:%d' % (vreg(D + 4 * k, 4), vreg(V_T + 9), 1024 * NW * k))
    b.e('s_mov_b64 exec, -1')
    b.e('s_waitcnt lgkmcnt(0)')
    for k in range(U):
        b.e('s_mov_b64 exec, %s' % sreg(78 + 2 * k, 2))
        b.e('global_store_dwordx4 %s, %s, %s' % (vreg(GOFF + k), vreg(D + 4 * k, 4), sreg(S_T + 4, 2)))
    b.e('s_mov_b64 exec, -1')
    b.e('v_add_u32_e32 %s, %d, %s' % (vreg(V_T + 9), 1024 * NW * U, vreg(V_T + 9)))
    b.e('s_add_u32 %s, %s, %d' % (sreg(S_T + 4), sreg(S_T + 4), 1024 * NW * U))
    b.e('s_addc_u32 %s, %s, 0' % (sreg(S_T + 5), sreg(S_T + 5)))
    b.e('s_add_u32 %s, %s, %d' % (sreg(S_T + 3), sreg(S_T + 3), 64 * NW * U))
    b.e('s_cmp_lt_u32 %s, %s' % (sreg(S_T + 3), sreg(S_T + 2)))
    b.e('s_cbranch_scc1 L_copy_%d' % h)
    b.e('s_branch L_end')


def epilogue(b, h, dbg=()):
    b.label('L_epilogue_%d' % h)
    b.e('s_waitcnt vmcnt(0)')                 # the re-fetched head of the stream has landed: the ring may go
    b.e('s_nop 15')
    b.e('s_nop 15')
    stamp(b, 2, 10 + h)
    # ---- range guard (hs_report_peak): pk1 = this lane's row maximum of |h1| hi halves, pk2 = of |h2| hi halves
    b.e('s_cmp_eq_u64 %s, 0' % sreg(S_PEAK, 2))
    b.e('s_cbranch_scc1 L_noguard_%d' % h)
    if 'noguard' in dbg:
        b.e('s_branch L_noguard_%d' % h)
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
        b.e('s_branch L_end')
    if 'rowstores' not in dbg:
        epilogue_staged(b, h, 'L_rowstores_%d' % h)
    b.label('L_rowstores_%d' % h)
    # ---- output: lane = row, register quad = 4 consecutive outputs (columns 128 h + 32 jj + 8 rq + 4 hi + e)
    b.e('s_mov_b64 exec, %s' % sreg(S_ROWMASK, 2))
    for jj in range(4):
        for rq in range(4):
            c0 = 128 * h + 32 * jj + 8 * rq                 # + 4 hi + e
            imm = (32 * jj + 8 * rq) * 4
            uid = 'L_o%d_%d_%d' % (h, jj, rq)
            b.e('ds_read_b128 %s, %s offset:%d' % (vreg(V_T + 4, 4), vreg(V_B2ADDR), imm))
            for e in range(4):
                b.e('v_accvgpr_read_b32 %s, %s' % (vreg(V_T + e), areg(ACC2 + 16 * jj + 4 * rq + e)))
            b.e('s_waitcnt lgkmcnt(0)')
            for e in range(4):
                b.e('v_fma_f32 %s, %s, %s, %s' % (vreg(V_T + e), vreg(V_T + e), sreg(S_AS2), vreg(V_T + 4 + e)))
            # whole chunk (both hi halves) inside n2 ?
            b.e('s_cmp_ge_u32 %s, %d' % (sreg(S_N2), c0 + 8))
            b.e('s_cbranch_scc0 %s_m' % uid)
            b.e('global_store_dwordx2 %s, %s, %s offset:%d' % (vreg(V_OUTOFF), vreg(V_T, 2), sreg(S_OUT, 2), imm))
            b.e('global_store_dwordx2 %s, %s, %s offset:%d' % (vreg(V_OUTOFF), vreg(V_T + 2, 2), sreg(S_OUT, 2), imm + 8))
            b.e('s_branch %s_d' % uid)
            b.label('%s_m' % uid)
            for e in range(4):
                b.e('s_sub_i32 %s, %s, %d' % (sreg(S_T), sreg(S_N2), c0 + e))          # column valid <=> 4 hi < n2 - c0 - e
                b.e('v_cmp_gt_i32_e32 vcc, %s, %s' % (sreg(S_T), vreg(V_4HI)))
                b.e('s_and_b64 exec, vcc, %s' % sreg(S_ROWMASK, 2))
                b.e('global_store_dword %s, %s, %s offset:%d' % (vreg(V_OUTOFF), vreg(V_T + e), sreg(S_OUT, 2), imm + 4 * e))
            b.e('s_mov_b64 exec, %s' % sreg(S_ROWMASK, 2))
            b.label('%s_d' % uid)
    b.e('s_mov_b64 exec, -1')
    stamp(b, 3, 20 + h)
    b.e('s_branch L_end')


def kernel(name, dbg=(), mode='hs'):
    out = ['.globl %s' % name, '.p2align 8', '.type %s,@function' % name, '%s:' % name]
    pre = Block('common')
    common_prologue(pre, dbg, mode)
    blocks = [pre]
    r0 = Role(0, dbg, mode).build()
    r1 = Role(1, dbg, mode).build()
    e0, e1 = Block('ep0'), Block('ep1')
    epilogue(e0, 0, dbg)
    epilogue(e1, 1, dbg)
    blocks += r0 + [e0] + r1 + [e1]
    end = Block('end')
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


DESCRIPTOR = '''
.rodata
.p2align 6
.amdhsa_kernel {name}
  .amdhsa_group_segment_fixed_size {lds}
  .amdhsa_private_segment_fixed_size 0
  .amdhsa_kernarg_size {karg}
  .amdhsa_user_sgpr_count 2
  .amdhsa_user_sgpr_kernarg_segment_ptr 1
  .amdhsa_system_sgpr_workgroup_id_x 1
  .amdhsa_system_sgpr_workgroup_id_y {idy}
  .amdhsa_system_vgpr_workitem_id 0
  .amdhsa_next_free_vgpr 256
  .amdhsa_next_free_sgpr 96
  .amdhsa_accum_offset 128
  .amdhsa_reserve_vcc 1
  .amdhsa_float_round_mode_32 0
  .amdhsa_float_round_mode_16_64 0
  .amdhsa_float_denorm_mode_32 3
  .amdhsa_float_denorm_mode_16_64 3
  .amdhsa_dx10_clamp 1
  .amdhsa_ieee_mode 1
  .amdhsa_tg_split 0
.end_amdhsa_kernel
.text
'''

META_KERNEL = '''  - .name: {name}
    .symbol: {name}.kd
    .kernarg_segment_size: {karg}
    .kernarg_segment_align: 8
    .group_segment_fixed_size: {lds}
    .private_segment_fixed_size: 0
    .wavefront_size: 64
    .sgpr_count: 102
    .vgpr_count: 256
    .agpr_count: 128
    .max_flat_workgroup_size: 512
    .args:
      - .offset: 0
        .size: {karg}
        .value_kind: by_value
'''

VARIANTS = [('csi_band8', ()), ('csi_band8_cs', ('colsplit',)), ('csi_band8_bf16_cs', ('bf16', 'colsplit')), ('csi_band8_nostage', ('nostage',)), ('csi_band8_nostage_noreq', ('nostage', 'noreq')), ('csi_band8_bf16', ('bf16',)), ('csi_band8_bf16_nostage', ('bf16', 'nostage')), ('csi_band8_bf16_nostage_noaside', ('bf16', 'nostage', 'noconv', 'noreq')), ('csi_band8_bf16_noconv', ('bf16', 'noconv')), ('csi_band8_bf16_noaside', ('bf16', 'noconv', 'noreq')),
            ('csi_band8_bf16_skeleton', ('bf16', 'noconv', 'noreq', 'nodma', 'noread')), ('csi_band8_bf16_nostagger', ('bf16', 'nostagger')), ('csi_band8_noconv', ('noconv',)), ('csi_band8_noreq', ('noreq',)),
            ('csi_band8_noaside', ('noconv', 'noreq')), ('csi_band8_skeleton', ('noconv', 'noreq', 'nodma', 'noread')),
            ('csi_band8_skeleton_rnd', ('noconv', 'noreq', 'nodma', 'noread', 'rnd')), ('csi_band8_skeleton_rnd_nobarrier', ('noconv', 'noreq', 'nodma', 'noread', 'rnd', 'nobarrier')),
            ('csi_band8_noaside_rnd', ('noconv', 'noreq', 'rnd')), ('csi_band8_p2first', ('p2first',)), ('csi_band8_prio1', ('prio1',)), ('csi_band8_prio0', ('prio0',)), ('csi_band8_p2first_noaside', ('p2first', 'noconv', 'noreq')),
            ('csi_band8_nobarrier', ('nobarrier',)), ('csi_band8_stagger', ('stagger',)), ('csi_band8_nointerleave', ('nointerleave',)), ('csi_band8_ownpieces', ('ownpieces',)), ('csi_band8_nodma', ('nodma',)), ('csi_band8_noread', ('noread',)),
            ('csi_band8_noaside_nodma', ('noconv', 'noreq', 'nodma')), ('csi_band8_noaside_noread', ('noconv', 'noreq', 'noread')), ('csi_band8_exit0', ('exit0',)), ('csi_band8_exit1', ('exit1',)),
            ('csi_band8_bf16_nobarrier', ('bf16', 'nobarrier')), ('csi_band8_bf16_noread', ('bf16', 'noread')), ('csi_band8_bf16_nodma', ('bf16', 'nodma')),
            ('csi_band8_bf16_skeleton_nobarrier', ('bf16', 'noconv', 'noreq', 'nodma', 'noread', 'nobarrier')), ('csi_band8_bf16_skeleton_rnd', ('bf16', 'noconv', 'noreq', 'nodma', 'noread', 'rnd')),
            ('csi_band8_bf16_noaside_noread', ('bf16', 'noconv', 'noreq', 'noread')), ('csi_band8_bf16_noaside_nodma', ('bf16', 'noconv', 'noreq', 'nodma')),
            ('csi_band8_rowstores', ('rowstores',)), ('csi_band8_exit2', ('exit2',)), ('csi_band8_exit2_noguard', ('exit2', 'noguard')), ('csi_band8_exit2_nostore', ('exit2', 'nostore')), ('csi_band8_dump', ('dump',)), ('csi_band8_dump0', ('dump0',))]


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else 'gemm_hs_band8_gfx950.s'
    only = sys.argv[2:] or None
    parts = ['.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', '.text']
    meta = []
    for name, dbg in VARIANTS:
        if only and name not in only:
            continue
        parts.append(kernel(name, dbg, 'bf16' if 'bf16' in dbg else 'hs'))
        cs = 'colsplit' in dbg
        karg = KARG_BYTES_CS if cs else KARG_BYTES
        parts.append(DESCRIPTOR.format(name=name, karg=karg, lds=LDS_BYTES, idy=1 if cs else 0))
        meta.append(META_KERNEL.format(name=name, karg=karg, lds=LDS_BYTES))
    parts.append('.amdgpu_metadata\n---\namdhsa.version: [1, 2]\namdhsa.target: amdgcn-amd-amdhsa--gfx950\namdhsa.kernels:\n' + ''.join(meta) + '...\n.end_amdgpu_metadata\n')
    with open(path, 'w') as f:
        f.write('\n'.join(parts))


if __name__ == '__main__':
    main()
