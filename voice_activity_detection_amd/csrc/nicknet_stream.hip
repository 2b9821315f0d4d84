// Conv + GRU window classifier, stride-1 windows: the trunk once per feature row (nicknet_device.h has the cell).

#include "nicknet_device.h"

namespace vad {
namespace nn {

// ---------------------------------------------------------------------------------------------
// Stride-1 windows over a packed (F, 24) feature matrix.
//
// Window w covers rows w .. w + 29.  The output of conv level l (1-based) at position p of window w depends on rows
// w + p - l .. w + p + l; where those all lie inside the window -- positions l .. 29 - l, the "interior" -- the
// zero padding plays no part and the value is a property of the absolute row r = w + p alone: R_l[r].  A wave
// therefore evaluates every level ONCE per row for the 48 rows w0 .. w0 + 47 that its 16 windows touch (three row
// tiles: column n of tile T is row w0 + 16 T + n; the taps r - 1 and r + 1 are the neighbouring columns, fetched by
// row rotations through DPP), instead of once per window and position, and then only the positions within l of a
// window's ends per window ("edges": l per side at level l, the columns being the windows again).  Per 16 windows
// that is 3 + (2 + 4 + 6 + 8) * (level's share) position evaluations of the trunk instead of 30 per level; the GRU
// steps, which depend on the window's own start, are unchanged.  At step p the GRU of window n reads the trunk's
// edge tile (p < levels or p > 29 - levels) or row w0 + n + p of R: the row tiles rotate by one lane per step, so
// lane n always finds its row in its own lane of one of two tiles.  Every value is produced by the same chain of
// instructions on the same operands as in nicknet_kernel (a column of an MFMA depends on that column of B only):
// the results are bit-identical.

template <int KS>
struct KVals {
    float v[KS];
};

template <int NU>
__device__ __forceinline__ KVals<tile_ksteps(NU)> kvals_of(const Tile& t) {
    KVals<tile_ksteps(NU)> r;
    tile_kvals<NU>(t, r.v);
    return r;
}

// every lane n of a 16-lane row takes the value of lane (n + O) % 16 of its row
template <int O>
__device__ __forceinline__ float rot16(float x) {
    if constexpr (O % 16 == 0) {
        return x;
    } else {
        constexpr int ctrl = 0x120 + (16 - O % 16);              // row_ror:(16 - O): lane n reads lane n - (16 - O)
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xF, 0xF, false));
    }
}

template <int O>
__device__ __forceinline__ Tile rot16(const Tile& t) {
    Tile r;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) r.b[b][i] = rot16<O>(t.b[b][i]);
    return r;
}

__device__ __forceinline__ Tile select_tile(bool first, const Tile& a, const Tile& b) {
    Tile r;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int i = 0; i < 4; ++i) r.b[blk][i] = first ? a.b[blk][i] : b.b[blk][i];
    return r;
}

// column n := row w0 + n + S of the row tiles
template <int S>
__device__ __forceinline__ Tile row_view(const Tile (&R)[3], int n) {
    constexpr int T = S / 16, O = S % 16;
    static_assert(S >= 0 && T + (O ? 1 : 0) <= 2, "the view must lie inside the three row tiles");
    if constexpr (O == 0) {
        return R[T];
    } else {
        return select_tile(n + O < 16, rot16<O>(R[T]), rot16<O>(R[T + 1]));
    }
}

// features of row `row` (clamped to the matrix) as this lane's k-step values
__device__ __forceinline__ KVals<kXk> feature_row(const float* __restrict__ feat, long row, long last_row, int q) {
    row = row < 0 ? 0 : (row > last_row ? last_row : row);
    KVals<kXk> r;
    load_x_row(r.v, feat + row * kF + kXk * q);
    return r;
}

// Level LI + 1 from level LI >= 1: the interior rows Rn and the edge tiles ELn[P] (position P) and ERn[J] (position
// 29 - J) from the same of the level below.
template <bool LARGE, int LI>
__device__ __forceinline__ void conv_level(const Tile (&Rp)[3], const Tile (&ELp)[LI], const Tile (&ERp)[LI], Tile (&Rn)[3],
                                           Tile (&ELn)[LI + 1], Tile (&ERn)[LI + 1], const f32x4* a_lane, const float* w, int q,
                                           int n) {
    using D = Dims<LARGE>;
    constexpr int NU = D::chan(LI);
    auto kv = [](const Tile& t) { return kvals_of<NU>(t); };
    // ---- edges first: the level below's edge tiles die here ----
    auto left = [&](auto pc) {
        constexpr int P = decltype(pc)::value;
        const Tile zero{};
        const Tile& m1 = P > 0 ? ELp[P > 0 ? P - 1 : 0] : zero;
        Tile c0, p1;
        if constexpr (P < LI) c0 = ELp[P]; else c0 = row_view<P>(Rp, n);
        if constexpr (P + 1 < LI) p1 = ELp[P + 1]; else p1 = row_view<P + 1>(Rp, n);
        ELn[P] = conv_layer<LARGE, LI>(a_lane, w, kv(m1).v, kv(c0).v, kv(p1).v, P > 0, true, q);
        NN_STAGE_FENCE;
    };
    auto right = [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        const Tile zero{};
        const Tile& p1 = J > 0 ? ERp[J > 0 ? J - 1 : 0] : zero;
        Tile c0, m1;
        if constexpr (J < LI) c0 = ERp[J]; else c0 = row_view<kT - 1 - J>(Rp, n);
        if constexpr (J + 1 < LI) m1 = ERp[J + 1]; else m1 = row_view<kT - 2 - J>(Rp, n);
        ERn[J] = conv_layer<LARGE, LI>(a_lane, w, kv(m1).v, kv(c0).v, kv(p1).v, true, J > 0, q);
        NN_STAGE_FENCE;
    };
    left(std::integral_constant<int, 0>{});
    right(std::integral_constant<int, 0>{});
    left(std::integral_constant<int, 1>{});
    right(std::integral_constant<int, 1>{});
    if constexpr (LI >= 2) {
        left(std::integral_constant<int, 2>{});
        right(std::integral_constant<int, 2>{});
    }
    if constexpr (LI >= 3) {
        left(std::integral_constant<int, 3>{});
        right(std::integral_constant<int, 3>{});
    }
    static_assert(LI >= 1 && LI <= 3, "levels 2 .. 4");
    // ---- interior: one evaluation per row; the taps are the neighbouring columns (the end columns of a tile take
    //      theirs from the next tile; the outermost two columns of the 48 have no neighbour and hold rows nobody reads) ----
#pragma unroll
    for (int T = 0; T < 3; ++T) {
        // column n - 1 / n + 1 (mod 16) of the tile; the end column's neighbour is the adjacent tile's other end
        Tile m1 = rot16<15>(Rp[T]), p1 = rot16<1>(Rp[T]);
        if (T > 0) m1 = select_tile(n == 0, rot16<15>(Rp[T > 0 ? T - 1 : 0]), m1);
        if (T < 2) p1 = select_tile(n == 15, rot16<1>(Rp[T < 2 ? T + 1 : 2]), p1);
        Rn[T] = conv_layer<LARGE, LI>(a_lane, w, kv(m1).v, kv(Rp[T]).v, kv(p1).v, true, true, q);
        NN_STAGE_FENCE;
    }
}

template <int NQ>
__device__ __forceinline__ void shift_down(Tile (&qv)[NQ]) {
#pragma unroll
    for (int i = 0; i + 1 < NQ; ++i) qv[i] = qv[i + 1];
}

template <bool LARGE>
__global__ __launch_bounds__(kWaves * 64) void nicknet_stream_kernel(const float* __restrict__ feat, long n_windows,
                                                                     const float* __restrict__ blob, float* __restrict__ p_out,
                                                                     uint8_t* __restrict__ dec_out, int mode, float threshold) {
    using D = Dims<LARGE>;
    constexpr NnLayout LY = nn_layout<LARGE>();
    constexpr int NC = D::nc, C = D::chan(NC), H = D::hidden;
    extern __shared__ __align__(16) float lds[];
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(blob);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int i = threadIdx.x; i < LY.lds / 4; i += kWaves * 64) dst[i] = src[i];
    }
    __syncthreads();                                             // the only barrier

    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, n = lane & 15;
    const f32x4* a_lane = reinterpret_cast<const f32x4*>(lds) + lane;
    const long n_tiles = (n_windows + kWaveWindows - 1) / kWaveWindows;
    const long last_row = n_windows + kT - 2;                    // the matrix has n_windows + 29 rows

    for (long tile = (long)blockIdx.x * kWaves + (threadIdx.x >> 6); tile < n_tiles; tile += (long)gridDim.x * kWaves) {
        const long w0 = tile * kWaveWindows;
        const long w = w0 + n;
        auto xrow = [&](long s) { return feature_row(feat, w + s, last_row, q); };

        // ---- level 1 from the feature rows ----
        Tile R1[3], EL1[1], ER1[1];
        {
            const KVals<kXk> none{};
            EL1[0] = conv_layer<LARGE, 0>(a_lane, lds, none.v, xrow(0).v, xrow(1).v, false, true, q);
            NN_STAGE_FENCE;
            ER1[0] = conv_layer<LARGE, 0>(a_lane, lds, xrow(kT - 2).v, xrow(kT - 1).v, none.v, true, false, q);
            NN_STAGE_FENCE;
#pragma unroll
            for (int T = 0; T < 3; ++T) {
                R1[T] = conv_layer<LARGE, 0>(a_lane, lds, xrow(16 * T - 1).v, xrow(16 * T).v, xrow(16 * T + 1).v, true, true, q);
                NN_STAGE_FENCE;
            }
        }
        Tile R2[3], EL2[2], ER2[2];
        conv_level<LARGE, 1>(R1, EL1, ER1, R2, EL2, ER2, a_lane, lds, q, n);
        Tile R[3], LQ[NC], RQ[NC];                               // the last level: rows, left and right edge tiles
        {
            Tile R3[3], EL3[3], ER3[3];
            conv_level<LARGE, 2>(R2, EL2, ER2, R3, EL3, ER3, a_lane, lds, q, n);
            if constexpr (NC > 3) {
                conv_level<LARGE, 3>(R3, EL3, ER3, R, LQ, RQ, a_lane, lds, q, n);
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i) { R[i] = R3[i]; LQ[i] = EL3[i]; RQ[i] = ER3[i]; }
            }
        }

        // ---- the GRU over the 30 positions ----
        Tile h[D::layers];
        SplitTile hs[D::layers];                                 // written by every step before it is read
#pragma unroll
        for (int l = 0; l < D::layers; ++l) h[l].b[0] = h[l].b[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int l = 0; l < D::layers; ++l) hs[l].hi = hs[l].mid = hs[l].lo = rnn::u32x4{0u, 0u, 0u, 0u};
#pragma unroll 1
        for (int p = 0; p < kT; ++p) {
            // R has been rotated p times: lane n holds column (n + p) % 16 of each tile; R[0] and R[1] are the two
            // tiles that rows w0 + n + p fall in
            Tile x;
            if (p < NC) {                                        // wave-uniform
                x = LQ[0];
                shift_down<NC>(LQ);
            } else if (p > kT - 1 - NC) {
                x = RQ[NC - 1];                                  // position 29 - J: J = NC - 1 first
#pragma unroll
                for (int i = NC - 1; i > 0; --i) RQ[i] = RQ[i - 1];
            } else {
                x = select_tile(n + (p & 15) < 16, R[0], R[1]);
            }
#pragma unroll
            for (int T = 0; T < 3; ++T) R[T] = rot16<1>(R[T]);
            if ((p & 15) == 15) {
                R[0] = R[1];
                R[1] = R[2];
            }
            NN_STAGE_FENCE;
            gru_layers<LARGE>(h, hs, x, a_lane, lds, p == 0, q);
        }
        embed_head<LARGE>(h, blob, lane, w, n_windows, p_out, dec_out, mode, threshold, nullptr);
    }
}

int launch_stream(bool large, const float* feat, long n_windows, const float* blob, float* p, uint8_t* dec, int mode,
                  float threshold, int blocks, hipStream_t st) {
    if (large)
        hipLaunchKernelGGL(nicknet_stream_kernel<true>, dim3((unsigned)blocks), dim3(kWaves * 64),
                           (size_t)nn_layout<true>().lds * sizeof(float), st, feat, n_windows, blob, p, dec, mode, threshold);
    else
        hipLaunchKernelGGL(nicknet_stream_kernel<false>, dim3((unsigned)blocks), dim3(kWaves * 64),
                           (size_t)nn_layout<false>().lds * sizeof(float), st, feat, n_windows, blob, p, dec, mode, threshold);
    VAD_HIP_CHECK(hipGetLastError());
    return VAD_OK;
}

int stream_attrs() {
    VAD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nicknet_stream_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, nn_layout<true>().lds * (int)sizeof(float)));
    VAD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nicknet_stream_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, nn_layout<false>().lds * (int)sizeof(float)));
    return VAD_OK;
}

}  // namespace nn
}  // namespace vad
