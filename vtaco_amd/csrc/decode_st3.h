// Slot-pipelined lattice decode (split-f16 layers), included by decode.hip in the split-f16 translation unit.
//
// Same work per wave as decode_fwd_staged2_kernel -- a 2 x 4 x 8 double brick = two 32-point MFMA column groups A and B,
// the gather staged through the wave's LDS image -- but the MLP is written as an explicit software pipeline instead of
// leaving the interleave to the compiler (which emitted the 64 relu/split instructions of a layer as one block with no
// MFMA in flight, then 12 MFMAs back to back with the VALU idle).  On gfx950 a wave's VALU instructions overlap the
// matrix pipe only when they sit BETWEEN that wave's own MFMAs in program order (an MFMA holds the SIMD's vector issue
// for 8 of its 32 cycles), so the stream here is: one MFMA, then one relu + hi/lo split of a register PAIR of the
// OTHER column group (4-5 VALU instructions), pinned by scheduling barriers.  A block (fc_0, fc_1, the next block's
// fc_c conditioning and the bias MFMA) is four slots; group B runs one slot behind group A:
//
//     slot 0:  MFMA  A.cond(k-step 0) x3, A.fc_0 x6 -> hid_A    | VALU  split relu(net_B) -> sB
//     slot 1:  MFMA  A.bias, A.cond(k-step 1) x3, B.fc_0 x6      | VALU  split relu(hid_A) -> sA
//     slot 2:  MFMA  B.bias, B.cond(k-step 0) x3, A.fc_1 x6      | VALU  split relu(hid_B) -> sB
//     slot 3:  MFMA  B.cond(k-step 1) x3, B.fc_1 x6 -> net_B     | VALU  split relu(net_A) -> sA  (next block)
//
// Every slot's MFMAs read what the previous slot's VALU wrote and vice versa; the conditioning / bias MFMAs depend on
// nothing the block computes and open each slot, so the first split instructions find the previous chain complete.
// Weight fragments are requested from LDS one slot ahead.  fc_p is one f16 MFMA per group (coordinates as hi + mid halves
// from a per-axis table, weights as hi + lo: vt_common.h) instead of two f32 MFMAs of 64 cycles each.
// The next double brick's footprint travels global -> LDS by LDS-DMA (global_load_lds_dwordx4 with per-lane source
// addresses, which also lay down the 144-byte row pitch): no registers held across the MLP, no ds_write pass.
#pragma once

namespace {

typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

// V selects the arithmetic of the dense layers and the relu + split that feeds them:
//   0, 1  split-f16 ("f16x3", decode_common.h): W x = W_lo x_hi + W_hi x_lo + W_hi x_hi on six f16 MFMAs; V = 0 forms x_lo with
//         v_fma_mixlo / mixhi_f16 (4 instructions per register pair; the mix-to-half forms do not issue in an MFMA's shadow:
//         tools/probe/issue_probe.hip), V = 1 with 2 x v_fma_mix_f32 + v_cvt_pkrtz_f16_f32 (5 instructions that do);
//   3     no MLP at all: the staged gather alone, the sampled features written out (vt_sample_grid over a lattice slab)
//   2     "f16f8": W_hi x_hi on two f16 MFMAs + ONE fp8 (e4m3) 32x32x64 MFMA for both correction products -- k-slots 0..15 of a lane
//         half pair W_lo 2^(11+SW) with x_hi 2^-SX, k-slots 16..31 W_hi 2^SW with x_lo 2^(11-SX), the MFMA's block scales undo the
//         shifts -- 128 matrix cycles per layer instead of 192 at 40 % less matrix-pipe energy (tools/probe/shape_probe.hip); the
//         corrections carry 4 significant bits each: logits within ~4e-5 of f32 on the golden decoder (tools/probe/decode_f8_emul.py).
constexpr int ST3_F8_SX = VT_F8_SX;                 // activations enter the fp8 copies as x 2^-SX (|x_hi| up to 448 * 2^SX, then saturating)
constexpr int ST3_F8_SW = VT_F8_SW;                 // weights as W 2^SW (vt_decoder_pack_f16f8)

template <int V> struct SpT {                       // one column group's split activations: k-step s -> 8 halves
    u32x4 hi[2], lo[2];
};
template <> struct SpT<2> {
    u32x4 hi[2];
    u32x8 q;                                        // fp8 operand: dwords 0-3 = x_hi copies (registers 4d .. 4d+3), 4-7 = x_lo copies
};

__device__ __forceinline__ f32x16 mfma_h(const u32x4 &a, const u32x4 &b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// both correction products of a layer: D += 2^(SX - SW - 11) A8 B8 (E8M0 block scales, bias 127, the same in every lane)
__device__ __forceinline__ f32x16 mfma_q(const u32x8 &a, const u32x8 &b, f32x16 c) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(__builtin_bit_cast(i32x8, a), __builtin_bit_cast(i32x8, b), c, 0, 0,
                                                           0, 127 - 11 - ST3_F8_SW, 0, 127 + ST3_F8_SX);
}

// relu + split of one register pair in two halves, so that the head of pair p + 1 sits between the instructions of pair p's
// tail (no dependent instruction directly behind its producer):
//   head: hi = v_pk_max_f16(v_cvt_pkrtz_f16_f32(a, b), 0)            [V = 2: + v_cvt_scalef32_pk_fp8_f16 of hi]
//   tail: the differences lo = v - hi (clamped to [0, 1]: the relu of lo), then their pack [V = 2: v_cvt_scalef32_pk_fp8_f32]
template <bool RELU>
__device__ __forceinline__ unsigned split_head(float a, float b) {
    const f16x2 zero = {(_Float16)0.0f, (_Float16)0.0f};
    f16x2 hp = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
    if (RELU) hp = __builtin_elementwise_max(hp, zero);
    return __builtin_bit_cast(unsigned, hp);
}
// tail, first part: the two differences (V = 0: mix-to-half, already packed in l0's bits)
template <int V, bool RELU>
__device__ __forceinline__ void split_tail_a(float a, float b, unsigned hw, float m1, float &l0, float &l1) {
    const f16x2 zero = {(_Float16)0.0f, (_Float16)0.0f}, one = {(_Float16)1.0f, (_Float16)1.0f};
    const f16x2 hp = __builtin_bit_cast(f16x2, hw);
    if constexpr (V == 0) {
        f16x2 lo;
        lo[0] = (_Float16)__builtin_fmaf((float)hp[0], m1, a);
        lo[1] = (_Float16)__builtin_fmaf((float)hp[1], m1, b);
        if (RELU) lo = __builtin_elementwise_min(__builtin_elementwise_max(lo, zero), one);
        l0 = __builtin_bit_cast(float, lo);
        l1 = 0.0f;
    } else {
        l0 = __builtin_fmaf((float)hp[0], m1, a);
        l1 = __builtin_fmaf((float)hp[1], m1, b);
        if (RELU) {
            l0 = __builtin_fminf(__builtin_fmaxf(l0, 0.0f), 1.0f);
            l1 = __builtin_fminf(__builtin_fmaxf(l1, 0.0f), 1.0f);
        }
    }
}
// two fp8 (e4m3) values v / scale into word (PAIR & 1) of a dword of the fp8 operand, the other word kept.  An even pair opens its
// dword: whatever the register holds is fine (the odd pair overwrites the other word), so no instruction initialises it.
template <int PAIR>
__device__ __forceinline__ i16x2 fp8_old(unsigned old) {
    if constexpr (PAIR & 1) return __builtin_bit_cast(i16x2, old);
    else { i16x2 o; asm volatile("" : "=v"(o)); return o; }
}
template <int PAIR>
__device__ __forceinline__ unsigned fp8_pair_f16(unsigned old, unsigned hw, float scale) {
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(fp8_old<PAIR>(old), __builtin_bit_cast(f16x2, hw), scale, (PAIR & 1) != 0));
}
template <int PAIR>
__device__ __forceinline__ unsigned fp8_pair_f32(unsigned old, float l0, float l1, float scale) {
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(fp8_old<PAIR>(old), l0, l1, scale, (PAIR & 1) != 0));
}
// pair p (0..7) of an accumulator <-> dword p & 3 of k-step p >> 2 (f16 operands), word p & 1 of dword p >> 1 (fp8 operand)
template <int V, bool RELU, int PAIR>
__device__ __forceinline__ void head_a(SpT<V> &s, const f32x16 &x) {                  // hi = relu(half(v))
    s.hi[PAIR >> 2][PAIR & 3] = split_head<RELU>(x[2 * PAIR], x[2 * PAIR + 1]);
}
template <int V, int PAIR>
__device__ __forceinline__ void head_b(SpT<V> &s) {                                    // V = 2: the fp8 copy of hi
    if constexpr (V == 2) s.q[PAIR >> 1] = fp8_pair_f16<PAIR>(s.q[PAIR >> 1], s.hi[PAIR >> 2][PAIR & 3], (float)(1 << ST3_F8_SX));
}
template <int V, int PAIR>
__device__ __forceinline__ void pack_to(SpT<V> &s, float l0, float l1) {
    if constexpr (V == 0) s.lo[PAIR >> 2][PAIR & 3] = __builtin_bit_cast(unsigned, l0);
    else if constexpr (V == 1) s.lo[PAIR >> 2][PAIR & 3] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l0, l1));
    else s.q[4 + (PAIR >> 1)] = fp8_pair_f32<PAIR>(s.q[4 + (PAIR >> 1)], l0, l1, 1.0f / (float)(1 << (11 - ST3_F8_SX)));
}
// the split of an accumulator as eight steps for the MFMA gaps: H0 H1 | T0 H2 | T1 H3 | T2 H4 | T3 H5 | T4 H6 | T5 H7 | T6 T7
// (H = half conversion + relu of a pair, T = its differences and their pack; V = 2: a pair's fp8 copy of hi one step behind its H).
// Inside a step no instruction sits behind its producer closer than the hardware wants (each miss is a wait state): this pair's two
// differences, the next pair's conversion and relu, the fp8 copy, the pack; scheduling barriers keep that order.
template <int V, bool RELU, int K>
__device__ __forceinline__ void split_step(SpT<V> &s, const f32x16 &x, float m1, unsigned &rmax) {
    if constexpr (K == 0) {
        head_a<V, RELU, 0>(s, x);
        head_a<V, RELU, 1>(s, x);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RELU) range_track(rmax, s.hi[0][0]);          // the range guard's sample (decode_common.h)
        head_b<V, 0>(s);
    } else if constexpr (K == 7) {
        float a0, a1, b0, b1;
        split_tail_a<V, RELU>(x[12], x[13], s.hi[1][2], m1, a0, a1);
        split_tail_a<V, RELU>(x[14], x[15], s.hi[1][3], m1, b0, b1);
        __builtin_amdgcn_sched_barrier(0);
        head_b<V, 7>(s);
        pack_to<V, 6>(s, a0, a1);
        pack_to<V, 7>(s, b0, b1);
    } else {
        constexpr int P = K - 1;
        float l0, l1;
        split_tail_a<V, RELU>(x[2 * P], x[2 * P + 1], s.hi[P >> 2][P & 3], m1, l0, l1);
        __builtin_amdgcn_sched_barrier(0);
        head_a<V, RELU, K + 1>(s, x);
        __builtin_amdgcn_sched_barrier(0);
        head_b<V, K>(s);
        pack_to<V, P>(s, l0, l1);
    }
}
template <int V, bool RELU>
__device__ __forceinline__ SpT<V> split_all(const f32x16 &x, float m1) {
    SpT<V> s;
    unsigned unused = 0;
    split_step<V, RELU, 0>(s, x, m1, unused); split_step<V, RELU, 1>(s, x, m1, unused); split_step<V, RELU, 2>(s, x, m1, unused);
    split_step<V, RELU, 3>(s, x, m1, unused); split_step<V, RELU, 4>(s, x, m1, unused); split_step<V, RELU, 5>(s, x, m1, unused);
    split_step<V, RELU, 6>(s, x, m1, unused); split_step<V, RELU, 7>(s, x, m1, unused);
    return s;
}

constexpr int ST3_THREADS = 512;                  // 8 waves per CU, two per SIMD
constexpr int ST3_CHUNKS = ST2_ROWS * (ST_ROW_BYTES / 16);            // 16-byte chunks of a wave's image, pad chunks included: 648
constexpr int ST3_PIECES = ST3_CHUNKS / 64;                           // whole 1-KiB LDS-DMA pieces: 10 (+ one of 8 lanes)
static_assert(ST3_CHUNKS - 64 * ST3_PIECES == 8, "the last LDS-DMA piece is eight lanes");

#define ST3_GAP() __builtin_amdgcn_sched_barrier(0)
#define ST3_M(acc, w, x) do { acc = mfma_h(w, x, acc); ST3_GAP(); } while (0)   /* the MFMA opens its gap */
#define ST3_Q(acc, w, x) do { acc = mfma_q(w, x, acc); ST3_GAP(); } while (0)   /* fp8 correction MFMA (V = 2) */
#define ST3_RUN_BLOCKS() do { _Pragma("unroll 1") for (int i = 0; i < 4; ++i) block(i, std::true_type{}); block(4, std::false_type{}); } while (0)
#define ST3_S(dst, src, k) split_step<V, true, k>(dst, src, m1, rmax)   /* step k of a relu + split */
#define ST3_C(dst, src, k) split_step<V, false, k>(dst, src, m1, rmax)  /* step k of a plain split (the sampled features) */

template <int V>
__global__ void __launch_bounds__(ST3_THREADS)
decode_fwd_staged3_kernel(DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ClockStamp stamp = clock_begin(a.clk);
    // V = 2: MODE.FP16_OVFL = 1 -- the fp8 conversions then saturate at +-448 instead of producing NaN (hwreg MODE = 1, bit 23)
    if constexpr (V == 2) __builtin_amdgcn_s_setreg((1 - 1) << 11 | 23 << 6 | 1, 1);
    if constexpr (V != 3) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.blob);
        f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
        for (int i = threadIdx.x; i < VT_BLOB_FLOATS / 4; i += ST3_THREADS) dst[i] = src[i];
    }
    const int R = a.R;
    AxisEnt *tab = reinterpret_cast<AxisEnt *>(lds + VT_BLOB_FLOATS);
    unsigned *ptab = reinterpret_cast<unsigned *>(tab + a.nx);            // coordinate of lattice index i as f16 hi | mid << 16
    for (int i = threadIdx.x; i < a.nx; i += ST3_THREADS) {
        float p, unused0, unused1;
        lattice_point(a, (uint32_t)i, 0u, 0u, p, unused0, unused1);
        const float f = grid_coord(p, a.divisor, R);
        const float f0 = floorf(f);
        AxisEnt e;
        e.i0 = (int)f0;
        e.w0 = (f0 + 1.0f) - f;
        e.i1 = min(e.i0 + 1, R - 1);
        e.w1 = (e.i0 + 1 <= R - 1) ? f - f0 : 0.0f;
        tab[i] = e;
        const _Float16 ph = (_Float16)p;
        const _Float16 pm = (_Float16)(p - (float)ph);
        ptab[i] = (unsigned)__builtin_bit_cast(unsigned short, ph) | ((unsigned)__builtin_bit_cast(unsigned short, pm) << 16);
    }
    if (threadIdx.x == 0)
        *reinterpret_cast<unsigned *>(reinterpret_cast<char *>(lds + VT_BLOB_FLOATS + 5 * a.nx) + (ST3_THREADS / 64) * ST2_WAVE_BYTES) = 0u;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int pl = lane & 31;
    const int h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int WPB = ST3_THREADS / 64;
    const uint32_t ntiles = a.total >> 6;                                // double bricks
    const uint32_t nx = (uint32_t)a.nx;
    const uint32_t tpb = a.N >> 6, q4 = nx >> 2, q8 = nx >> 3;
    const uint32_t plane0 = a.lattice_first / (nx * nx);
    char *stage = reinterpret_cast<char *>(lds + VT_BLOB_FLOATS + 5 * a.nx) + wave * ST2_WAVE_BYTES;
    // the workgroup's tile counter (behind the images): waves CLAIM their tiles, see below
    unsigned *claim_ctr = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(lds + VT_BLOB_FLOATS + 5 * a.nx) + WPB * ST2_WAVE_BYTES);
    // LDS byte address of the wave's image (M0 of the LDS-DMA pieces): the low half of the generic pointer
    const unsigned stage_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)reinterpret_cast<size_t>(stage));
    const unsigned stage_lds_last = (unsigned)__builtin_amdgcn_readfirstlane((int)(stage_lds + 1024u * ST3_PIECES));

    // LDS-DMA piece k writes the 16-byte chunks 64k .. 64k+63 of the image [72 rows][9 chunks] (chunk 8 of a row is the
    // 16-byte pad: it receives the row's chunk 0 again); row = (dz * 4 + dy) * 3 + dx of the 6 x 4 x 3 voxel footprint
    uint32_t src_off[ST3_PIECES + 1];
#pragma unroll
    for (int k = 0; k <= ST3_PIECES; ++k) {
        const int m = min(64 * k + lane, ST3_CHUNKS - 1), row = m / 9, c = m - 9 * row;
        const int dx = row % 3, t = row / 3, dy = t & 3, dz = t >> 2;
        src_off[k] = (uint32_t)((dz * R + dy) * R + dx) * 128u + (uint32_t)((c == 8) ? 0 : c) * 16u;
    }

    uint32_t t_begin = 0, t_end = ntiles, w_idx = blockIdx.x * WPB + wave, w_cnt = gridDim.x * WPB;
    if ((gridDim.x & 7u) == 0 && ntiles >= 8u * WPB) {                  // XCD-aware order, as decode_fwd_kernel
        const uint32_t chunk = (ntiles + 7u) >> 3, xcd = blockIdx.x & 7u;
        t_begin = min(xcd * chunk, ntiles);
        t_end = min(t_begin + chunk, ntiles);
        w_idx = (blockIdx.x >> 3) * WPB + wave;
        w_cnt = (gridDim.x >> 3) * WPB;
    }
    auto brick_of = [&](uint32_t tile, uint32_t &b, uint32_t &X0, uint32_t &Y0, uint32_t &Z0) {
        b = tile / tpb;
        const uint32_t t = tile - b * tpb;
        const uint32_t pp = t / (q4 * q8), rem = t - pp * q4 * q8;
        const uint32_t by = rem / q8, bz = rem - by * q8;
        X0 = 2u * pp; Y0 = 4u * by; Z0 = 8u * bz;
    };
    // footprint origin (clamped so that the 3 x 4 x 6 block stays inside the grid) and its eleven LDS-DMA pieces.  The
    // pieces are invisible to the compiler: the loop waits for them (vmcnt) before the gather reads the image.
    int ox = 0, oy = 0, oz = 0;
    auto fetch = [&](uint32_t tile, int &fx, int &fy, int &fz) {
        uint32_t b, X0, Y0, Z0;
        brick_of(tile, b, X0, Y0, Z0);
        fx = min(__builtin_amdgcn_readfirstlane(tab[plane0 + X0].i0), R - 3);
        fy = min(__builtin_amdgcn_readfirstlane(tab[Y0].i0), R - 4);
        fz = min(__builtin_amdgcn_readfirstlane(tab[Z0].i0), R - 6);
        const uint64_t bp = reinterpret_cast<uint64_t>(a.grid + ((((size_t)b * R + fz) * R + fy) * R + fx) * 32);
        const char *base = reinterpret_cast<const char *>(
            ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bp >> 32)) << 32) |
            (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bp));
        unsigned keep;
        asm volatile(
            "s_mov_b32 %[keep], m0\n\t"
            "s_nop 4\n\t"
            "s_mov_b32 m0, %[dst]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o0], %[base]\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o1], %[base]\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o2], %[base]\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o3], %[base]\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o4], %[base]\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o5], %[base]\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o6], %[base]\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o7], %[base]\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o8], %[base]\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o9], %[base]\n\t"
            "s_mov_b32 m0, %[keep]"
            : [keep] "=&s"(keep)
            : [dst] "s"(stage_lds), [base] "s"(base), [o0] "v"(src_off[0]), [o1] "v"(src_off[1]), [o2] "v"(src_off[2]),
              [o3] "v"(src_off[3]), [o4] "v"(src_off[4]), [o5] "v"(src_off[5]), [o6] "v"(src_off[6]), [o7] "v"(src_off[7]),
              [o8] "v"(src_off[8]), [o9] "v"(src_off[9])
            : "memory", "scc");
        if (lane < ST3_CHUNKS - 64 * ST3_PIECES) {
            asm volatile(
                "s_mov_b32 %[keep], m0\n\t"
                "s_nop 4\n\t"
                "s_mov_b32 m0, %[dst]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o], %[base]\n\t"
                "s_mov_b32 m0, %[keep]"
                : [keep] "=&s"(keep)
                : [dst] "s"(stage_lds_last), [base] "s"(base), [o] "v"(src_off[ST3_PIECES])
                : "memory");
        }
    };

    // B operand of the bias MFMA: ones in the three k-slots that carry the bias's hi / mid / lo parts
    u32x4 ones;
    {
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)((h == 0 && e < 3) ? 1.0f : 0.0f);
        ones = __builtin_bit_cast(u32x4, o);
    }
    const float m1 = opaque_minus_one();
    unsigned rmax = 0;                                                   // range guard: largest sampled hi half of this wave
    [[maybe_unused]] float lmax = 0.0f;                                  // V = 2: largest |logit| this lane wrote (VT_RANGE_LOGIT)

    // Tiles are CLAIMED, not pre-assigned.  The two waves that share a SIMD do not share it evenly: VALU issue is arbitrated by
    // age, so with a fixed 16 tiles each the older wave (0-3) was done after ~70 % of the launch and its partner ran the rest
    // alone -- without the other wave's MFMAs to overlap its relu / split with (tools/diag_wg.py: workgroup 0's first wave
    // 114 us of a 164 us launch).  The workgroup owns the same set of tiles as before (tile i of it = the i-th of the strided
    // list below); a wave takes the next one from an LDS counter when it issues its prefetch, so the waves end within a tile of
    // each other.  Which wave computes a tile changes nothing in its result.
    const uint32_t wg_first = t_begin + w_idx - (uint32_t)wave;          // this workgroup's first tile
    unsigned fixed_next = (unsigned)wave;
    auto claim = [&]() -> uint32_t {
        unsigned i = 0;
        if (!a.claim) { i = fixed_next; fixed_next += (unsigned)WPB; }                      // A/B: the fixed share (wave w: w, w + WPB, ...)
        else {
            if (lane == 0) i = __hip_atomic_fetch_add(claim_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            i = (unsigned)__builtin_amdgcn_readfirstlane((int)i);
        }
        const uint32_t t = wg_first + (i % (unsigned)WPB) + (i / (unsigned)WPB) * w_cnt;
        return i < 0x10000u ? t : t_end;                                 // (the list is a few dozen tiles long)
    };
    uint32_t tile = claim();
    if (tile < t_end) fetch(tile, ox, oy, oz);
    uint32_t next_tile = t_end;
    for (; tile < t_end; tile = next_tile) {
        unsigned lds_off = 0;
        asm volatile("" : "+v"(lds_off));                                // see decode_fwd_kernel
        const float *L = lds + lds_off;

        uint32_t b, X0, Y0, Z0;
        brick_of(tile, b, X0, Y0, Z0);
        const uint32_t ixl = X0 + (uint32_t)(pl >> 4), iy = Y0 + (uint32_t)((pl >> 2) & 3), izA = Z0 + (uint32_t)(pl & 3), izB = izA + 4u;
        const uint32_t gA = b * a.N + (ixl * nx + iy) * nx + izA, gB = gA + 4u;
        const AxisEnt ex = tab[plane0 + ixl], ey = tab[iy], ezA = tab[izA], ezB = tab[izB];
        // LDS byte addresses of the four (y, x) corner columns of this lane's point, shared by both groups and both z planes
        // (32-bit LDS address arithmetic: generic-pointer arithmetic made every corner a 64-bit multiply-add)
        const unsigned img_lds = stage_lds + 64u * (unsigned)h;
        const unsigned cx0 = (unsigned)(ex.i0 - ox) * ST_ROW_BYTES, cx1 = (unsigned)(ex.i1 - ox) * ST_ROW_BYTES;
        const unsigned cy0 = (unsigned)(ey.i0 - oy) * (3u * ST_ROW_BYTES) + img_lds, cy1 = (unsigned)(ey.i1 - oy) * (3u * ST_ROW_BYTES) + img_lds;
        const unsigned c00 = cy0 + cx0, c01 = cy0 + cx1, c10 = cy1 + cx0, c11 = cy1 + cx1;
        const float w00 = ex.w0 * ey.w0, w01 = ex.w1 * ey.w0, w10 = ex.w0 * ey.w1, w11 = ex.w1 * ey.w1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this tile's footprint has landed in the image
        typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
        auto corner = [&](unsigned addr) {
            lds_f32x4 *q = reinterpret_cast<lds_f32x4 *>(addr);
            const f32x4 a = q[0], b = q[1], c = q[2], d = q[3];
            f32x16 r;
            r.s0 = a.x; r.s1 = a.y; r.s2 = a.z; r.s3 = a.w;
            r.s4 = b.x; r.s5 = b.y; r.s6 = b.z; r.s7 = b.w;
            r.s8 = c.x; r.s9 = c.y; r.sa = c.z; r.sb = c.w;
            r.sc = d.x; r.sd = d.y; r.se = d.z; r.sf = d.w;
            return r;
        };
        auto gather = [&](const AxisEnt &ez) {
            f32x16 c;
#pragma unroll
            for (int s = 0; s < 16; ++s) c[s] = 0.0f;
            auto plane = [&](unsigned cz, float wz) {                     // same corner and FMA order as the other decode kernels
                const f32x16 v00 = corner(cz + c00), v01 = corner(cz + c01);
                const float a0 = w00 * wz, a1 = w01 * wz;
#pragma unroll
                for (int s = 0; s < 16; ++s) c[s] = fmaf(v00[s], a0, c[s]);
#pragma unroll
                for (int s = 0; s < 16; ++s) c[s] = fmaf(v01[s], a1, c[s]);
                const f32x16 v10 = corner(cz + c10), v11 = corner(cz + c11);
                const float a2 = w10 * wz, a3 = w11 * wz;
#pragma unroll
                for (int s = 0; s < 16; ++s) c[s] = fmaf(v10[s], a2, c[s]);
#pragma unroll
                for (int s = 0; s < 16; ++s) c[s] = fmaf(v11[s], a3, c[s]);
            };
            plane((unsigned)(ez.i0 - oz) * (12u * ST_ROW_BYTES), ez.w0);
            pin16(c);
            __builtin_amdgcn_sched_barrier(0);
            plane((unsigned)(ez.i1 - oz) * (12u * ST_ROW_BYTES), ez.w1);
            pin16(c);
            __builtin_amdgcn_sched_barrier(0);
            return c;
        };
        const f32x16 cA = gather(ezA);
        const f32x16 cB = gather(ezB);
        // every read of the image has returned (the FMAs above consumed it): the next footprint may overwrite it
        next_tile = claim();
        if (next_tile < t_end) fetch(next_tile, ox, oy, oz);

        if constexpr (V == 3) {                                         // sample only: a.out is [points of the slab][32]
            store_gather16(a.out + (size_t)gA * 32, cA, h);
            store_gather16(a.out + (size_t)gB * 32, cB, h);
        } else {
        // ---- fc_p operands: the point's coordinates as half pairs from the per-axis table ----
        u32x4 pA, pB;
        {
            const unsigned tx = ptab[plane0 + ixl], ty = ptab[iy], tzA = ptab[izA], tzB = ptab[izB];
            const unsigned xy = (tx & 0xffffu) | (ty << 16);
            pA = h ? u32x4{tzA & 0xffffu, 0u, 0u, 0u} : u32x4{tx, ty, tzA, xy};
            pB = h ? u32x4{tzB & 0xffffu, 0u, 0u, 0u} : u32x4{tx, ty, tzB, xy};
        }
        const u32x4 *W = reinterpret_cast<const u32x4 *>(L);             // 16-byte fragments: [float offset / 4 + lane]
        auto frag = [&](int float_off) { return W[float_off / 4 + lane]; };
        [[maybe_unused]] auto frag8 = [&](int float_off) {                // two fragments = the eight dwords of an fp8 operand
            const u32x4 lo4 = W[float_off / 4 + lane], hi4 = W[float_off / 4 + 64 + lane];
            return u32x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        };

        f32x16 netA = load_frag16(L + VT_OFF_BIAS + h * 16);
        f32x16 netB = netA;
        SpT<V> csA, csB, sA, sB;

        // ---- optional tactile concat (fc_p_img's c_img columns): plain, not pipelined ----
        if (a.cimg_ids || a.c_img) {
            f32x16 ciA, ciB;
            bool has_img = true;
            if (a.cimg_ids) {
                const unsigned idA = a.cimg_ids[gA], idB = a.cimg_ids[gB];
                has_img = __ballot(idA < a.cimg_nf || idB < a.cimg_nf) != 0ull;  // most double bricks touch no finger
#pragma unroll
                for (int s = 0; s < 16; ++s) { ciA[s] = 0.0f; ciB[s] = 0.0f; }
                if (has_img && idA < a.cimg_nf) ciA = load_frag16(a.cimg_table + (size_t)idA * 32 + 16 * h);
                if (has_img && idB < a.cimg_nf) ciB = load_frag16(a.cimg_table + (size_t)idB * 32 + 16 * h);
            } else {
                ciA = load_frag16(a.c_img + (size_t)gA * 32 + 16 * h);
                ciB = load_frag16(a.c_img + (size_t)gB * 32 + 16 * h);
            }
            if (has_img) {
                if constexpr (V == 2) {
                    const SpT<2> iA = split_all<2, false>(ciA, m1), iB = split_all<2, false>(ciB, m1);
                    const u32x4 ih0 = frag(VT_OFF_WPI), ih1 = frag(VT_OFF_WPI + 256);
                    const u32x8 iq = frag8(VT_OFF_WPI + 512);
                    netA = mfma_h(ih0, iA.hi[0], netA); netB = mfma_h(ih0, iB.hi[0], netB);
                    netA = mfma_h(ih1, iA.hi[1], netA); netB = mfma_h(ih1, iB.hi[1], netB);
                    netA = mfma_q(iq, iA.q, netA); netB = mfma_q(iq, iB.q, netB);
                } else {
                    dense32s2<2>(netA, netB, L + VT_OFF_WPI, split16<false, 2>(ciA), split16<false, 2>(ciB), lane);
                }
            }
        }

        // weight fragments in flight: w0 = fc_0, w1 = fc_1 of the current block, c = fc_c of the next block (h = hi part of k-step
        // 0 / 1; o = lo part (V = 0, 1); q = fp8 correction fragment (V = 2)), bf = the block-end bias fragment
        u32x4 w0h0, w0h1, w1h0, w1h1, ch0, ch1, bf;
        [[maybe_unused]] u32x4 w0o0, w0o1, w1o0, w1o1, co0, co1;
        [[maybe_unused]] u32x8 w0q, w1q, cq;
        f32x16 hidA, hidB;
        // the pipeline itself -- prologue and the `block` lambda -- is written out by gen_st3.py
        if constexpr (V == 2) {
#include "decode_st3_f16f8.inc"
            ST3_RUN_BLOCKS();
        } else {
#include "decode_st3_f16x3.inc"
            ST3_RUN_BLOCKS();
        }
        }                                                                // V != 3
    }
    range_report(rmax, a.status, V == 2);
    if constexpr (V == 2) {
        if (a.status != nullptr && lmax > VT_F16F8_LOGIT_LIMIT) atomicOr(a.status, VT_RANGE_LOGIT);
    }
    clock_end(a.clk, stamp, reinterpret_cast<unsigned long long *>(claim_ctr + 4));
}

#undef ST3_GAP
#undef ST3_M
#undef ST3_S
#undef ST3_C

}  // namespace
