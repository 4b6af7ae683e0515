// Halo-patch kernel of the split mode for the STRIDE-2 data gradients / transposed convs of the headline path: the folded ConvMeanPool /
// UpsampleConv filters (4x4, stride 2, SAME = pad 1: TF/CT_gan_cifar_resnet.py:89-107 as one strided conv, functional.py).  A fragment of
// igemm16.hip: included inside its anonymous namespace, after conv16x3hf_kernel, whose machinery (P16, PatchGeom, the fragment-order
// filter image, the epilogue conventions) it shares.
//
// Why it exists: on the 32-deep slice kernel (conv16_kernel<3,2,*,32>) these layers ran at 122-162 TFLOP/s - every tap of every 32-channel
// chunk staged its own copy of the pixel operand (split VALU + LDS stores + two barriers per 48 MFMAs).  The halo form stages a patch
// once per chunk and runs EVERY tap from it; for stride 2 that takes the polyphase view of the filter: the output pixels of parity
// (a,b) are a stride-1 correlation of dy with the 2x2 taps of that parity (phase_geom), and all four phases read the SAME 3x3-halo patch
// of dy.  One workgroup = TN*32 positions of the dy grid x 128 output channels x FOUR phases: 16 (phase, tap) steps per staged patch,
// where one phase per workgroup gave 4 (measured slower than the slice kernel in round 3, DESIGN_HISTORY 4.7).  Measured (tools/
// conv16_bench.py f32x3 s2): 155-196 TFLOP/s against 122-162.
// Filter operand: fragment-order image, 16 steps of 6 KB per (32-channel block, 32-channel chunk), streamed from L2 one step ahead
// (frag_u32_index with RS = 16, step = 4*phase + tap of the phase).  EIGHT waves per workgroup, one workgroup per CU at 64-position
// tiles: a wave owns two phases, so four phases cost it no more accumulators than a stride-1 tile, and launches of 512 / 768 / 1280
// tiles run in whole rounds of 256.
// Arithmetic: chunk-major, then step, k step, the six products small-first into fp32 accumulators - per phase the same (chunk, tap)
// order as the slice kernel's walk.
// The FORWARD of the same filters stays on the slice kernel: a forward pixel of x feeds only 4 of the 16 taps (a dy pixel feeds all 16
// steps here), so a de-interleaved four-plane patch is 5.3 staged pixels per output pixel and the split VALU of its staging, not the
// matrix pipe, bounds it - built and measured in round 4 (eight waves, tap halves summed through LDS): 128-139 TFLOP/s on 16-wide
// outputs against the slice kernel's 128-146, 121-150 on 8-wide outputs against 106-113 but a net loss in the step (13.82 against
// 13.61 ms per iteration with the data gradient alone) - removed.

template <class F, int... I>
__device__ __forceinline__ void s2_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void s2_static_for(F&& f) { s2_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ---- data gradient: PatchGeom {TR rows of the dy grid per tile, PW = Q + 2, NPX = (TR + 2) * PW, n_it}
// EIGHT waves: wave = (32-channel block kb = wave & 3, row parity a = wave >> 2) owns the two phases (a, 0), (a, 1) of its channels -
// 2 * TN accumulators (four phases per wave would need 128 accumulator registers beside two filter-fragment sets: spills at two
// waves per SIMD).  All eight share the staged patch; a wave runs 8 of the 16 steps of a chunk.
template <int TN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(TN == 1 ? 4 : 2))) void conv16x3p_kernel(const P16 p, const PatchGeom pg) {
    constexpr int MMA = CTGAN_MMA_F32X3, NP = 3, BK = 32, NT = 512, MAXIT = TN == 2 ? 3 : 2, BMP = TN * 32;
    constexpr int LDS_K = BK + 8;
    constexpr int LDE = 32 + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    const int PPLANE = pg.NPX * LDS_K;
    unsigned short* const Xs = smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = wave & 3, pa = wave >> 2;
    const int tiles_n = p.Ng / 128;
    int bid = blockIdx.x;
    const int nb = gridDim.x;
    if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);       // neighbouring tiles (shared halo rows) on one XCD
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * BMP, n0 = tile_n * 128;
    const int nch = p.C / BK;
    const int PQ = p.P * p.Q;
    const int img = m0 / PQ, row0 = (m0 - img * PQ) / p.Q;

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t f_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.Wf), 0, p.wf_bytes, 0x00020000);
    const unsigned a_voff = (unsigned)lane * 16u;
    // this wave's filter stream: steps 8 * pa .. 8 * pa + 7 of every chunk of its 32-channel block (16 steps of 6 KB per chunk)
    unsigned a_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(((long long)((n0 >> 5) + kb) * nch * 16 + 8 * pa) * 6144));
    constexpr int DIST = TN == 2 ? 2 : 1, NSET = 2 * DIST;      // filter fragments DIST steps ahead (the 32-position tile runs four waves per SIMD on 128 registers: one)
    u32x4 fa[NSET][2][NP];                                // [register set][k step][plane]
    auto loadA = [&](auto setc, unsigned soff) __attribute__((always_inline)) {
        constexpr int SET = decltype(setc)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int q = 0; q < NP; ++q)
                fa[SET][ks][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(f_rsrc, a_voff, soff + (unsigned)((ks * NP + q) * 1024), 0));
    };
    // ---- patch loader: dy rows row0 - 1 .. row0 + TR, columns -1 .. Q (rows / columns outside dy: the descriptor's range check gives zeros)
    float4 rp[MAXIT];
    unsigned p_voff[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int item = it * NT + tid, px = item >> 3;
        p_voff[it] = 0xFFFFFFFFu;
        if (it < pg.n_it && px < pg.NPX) {
            const int prow = px / pg.PW, pcol = px - prow * pg.PW;
            const int ih = row0 + prow - 1, iw = pcol - 1;
            if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                p_voff[it] = (unsigned)(((long long)img * p.s_n + (long long)ih * p.s_h + (long long)iw * p.s_w + (item & 7) * 4) * 4);
        }
    }
    auto load_patch = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < MAXIT; ++it)
            if (it < pg.n_it)
                rp[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, p_voff[it] == 0xFFFFFFFFu ? 0xFFFFFFFFu : p_voff[it] + chunk * (BK * 4), 0, 0));
    };
    auto store_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int item = it * NT + tid, px = item >> 3;
            if (it < pg.n_it && px < pg.NPX) {
                const float4 v = rp[it];
                unsigned o0[NP], o1[NP];
                split_pk<MMA>(v.x, v.y, o0);
                split_pk<MMA>(v.z, v.w, o1);
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const u32x2 o = {o0[q], o1[q]};
                    *reinterpret_cast<u32x2*>(&Xs[q * PPLANE + px * LDS_K + (item & 7) * 4]) = o;
                }
            }
        }
    };

    f32x16 acc[2][TN];                                     // [column parity b][32-position sub-tile]
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[b][j][e] = 0.f;
    const int h = lane >> 5, l31 = lane & 31;
    int pix[TN];                                           // patch element offset of this lane's position (patch (0, 0) = dy (row0 - 1, -1)), row parity included
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int tp = j * 32 + l31;
        pix[j] = ((tp / p.Q + pa) * pg.PW + (tp % p.Q)) * LDS_K + h * 8;
    }
    constexpr int QW[6] = {2, 0, 1, 1, 0, 0}, QX[6] = {0, 2, 1, 0, 1, 0};      // (filter piece, pixel piece): l*h, h*l, m*m, m*h, h*m, h*h
    using set0 = std::integral_constant<int, 0>;

    load_patch(0);
    loadA(set0{}, a_base);
    if constexpr (DIST == 2) loadA(std::integral_constant<int, 1>{}, a_base + 6144u);
    for (int c = 0; c < nch; ++c) {
        store_patch();
        if (c + 1 < nch) load_patch(c + 1);                // in flight during this chunk's steps
        __syncthreads();
        const bool more = c + 1 < nch;
        // step S = 4 * b + 2 * t + u of phase (a, b): dy (i - pad_t[a] + t, j - pad_l[b] + u), pad_t = pad_l = {1, 0} (phase_geom of
        // R = S = 4, pad 1) = patch (i - row0 + a + t, j + b + u)
        s2_static_for<8>([&](auto sc) __attribute__((always_inline)) {
            constexpr int S = decltype(sc)::value, B = S >> 2, T = (S >> 1) & 1, U = S & 1, CUR = S % NSET, NXT = (S + DIST) % NSET;
            // (one step ahead left the waves parked 49 % of the time - PMC, profiles/r04_pmc_traffic_x3.json: a step is 24 MFMAs = 0.4 us,
            // less than an L2 round trip under load)
            if (S + DIST < 8) loadA(std::integral_constant<int, NXT>{}, a_base + (unsigned)((S + DIST) * 6144));
            else if (more) loadA(std::integral_constant<int, NXT>{}, a_base + (unsigned)((16 + S + DIST - 8) * 6144));
            const int tap_off = (T * pg.PW + (B + U)) * LDS_K;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 fx[NP][TN];
#pragma unroll
                for (int q = 0; q < NP; ++q)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        fx[q][j] = *reinterpret_cast<const u32x4*>(&Xs[q * PPLANE + pix[j] + tap_off + ks * 16]);
#pragma unroll
                for (int cl = 0; cl < 6; ++cl)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[B][j] = Cvt<MMA>::mma(fa[CUR][ks][QW[cl]], fx[QX[cl]][j], acc[B][j]);
            }
        });
        a_base += 16u * 6144u;
        __syncthreads();                                   // the patch may be overwritten
    }

    // epilogue through LDS, per phase: a wave's 32 channels x 32 positions per pass, transposed so that a lane stores 4 consecutive channels
    // of one output pixel (2i + a, 2j + b); bias / mask / residual / ReLU as the other kernels of the family
    float* es = reinterpret_cast<float*>(smem) + wave * (32 * LDE);
    constexpr int C4 = 8, ROWS_PER = 64 / C4;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const long long ph_off = (long long)pa * p.ph_d_h + (long long)b * p.ph_d_w;
#pragma unroll
        for (int jh = 0; jh < TN; ++jh) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = {acc[b][jh][4 * q], acc[b][jh][4 * q + 1], acc[b][jh][4 * q + 2], acc[b][jh][4 * q + 3]};
                *reinterpret_cast<float4*>(&es[l31 * LDE + 8 * q + 4 * h]) = v;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 32 / ROWS_PER; ++it) {
                const int row = it * ROWS_PER + lane / C4, c4 = lane % C4;
                const int m = m0 + jh * 32 + row, col = n0 + kb * 32 + c4 * 4;
                float4 v = *reinterpret_cast<const float4*>(&es[row * LDE + c4 * 4]);
                const int n = m / PQ, rem = m - n * PQ, pp = rem / p.Q, qq = rem - pp * p.Q;
                const long long off = n * p.ds_n + pp * p.ds_p + qq * p.ds_q + ph_off + col;
                if (p.bias) { const float4 bv = *reinterpret_cast<const float4*>(p.bias + col); v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
                if (p.mask) {
                    const float4 k = *reinterpret_cast<const float4*>(p.mask + off);
                    v.x = k.x > 0.f ? v.x : 0.f; v.y = k.y > 0.f ? v.y : 0.f; v.z = k.z > 0.f ? v.z : 0.f; v.w = k.w > 0.f ? v.w : 0.f;
                }
                if (p.resid) {
                    const float4 r = *reinterpret_cast<const float4*>(p.resid + off);
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4*>(p.D + off) = v;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// The data gradient of the folded 4x4 / stride-2 / pad-1 filters on conv16x3p_kernel: four phases of 2x2 taps with pads {1, 0}, gather
// stride 1 over dy, 128-channel tiles on the output side, 32-channel chunks, the phase-step fragment image, tiles of whole dy-grid rows
// inside one image.  bmp = positions of the dy grid per tile (64 / 32).
bool conv16x3p_ok(const P16& p, PatchGeom* out, int bmp) {
    if (p.nph != 4 || p.stride != 1 || p.Wf == nullptr || p.Ng % 128 || p.C % 32 || p.Q <= 0 || p.M % bmp) return false;
    if (p.ph_T[0] != 2 || p.ph_T[1] != 2 || p.ph_U[0] != 2 || p.ph_U[1] != 2) return false;
    if (p.ph_pad_t[0] != 1 || p.ph_pad_t[1] != 0 || p.ph_pad_l[0] != 1 || p.ph_pad_l[1] != 0) return false;
    if (p.H != p.P || p.W != p.Q || p.drop) return false;
    const int PQ = p.P * p.Q;
    if (PQ % bmp || bmp % p.Q) return false;
    PatchGeom g;
    g.IMGS = 1; g.TR = bmp / p.Q; g.PW = p.Q + 2; g.PIMG = (g.TR + 2) * g.PW; g.NPX = g.PIMG;
    g.n_it = (g.NPX * 8 + 511) / 512;
    if (g.n_it > (bmp == 64 ? 3 : 2)) return false;
    if (out) *out = g;
    return true;
}
// 64-position tiles (one workgroup of eight waves per CU) when they fill the chip, else 32-position tiles (two per CU)
int conv16x3p_tile(const P16& p) {
    const bool ok64 = conv16x3p_ok(p, nullptr, 64), ok32 = conv16x3p_ok(p, nullptr, 32);
    if (ok64 && (!ok32 || (long long)(p.M / 64) * (p.Ng / 128) >= 256)) return 64;
    return ok32 ? 32 : 0;
}

template <int TN>
int launch_conv16x3p_t(const P16& p, const PatchGeom& pg, hipStream_t st) {
    const size_t epi = (size_t)8 * 32 * 36 * 4, stage = (size_t)3 * pg.NPX * 40 * 2;
    const size_t lds = stage > epi ? stage : epi;
    static size_t have = 0;
    if (have < lds) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv16x3p_kernel<TN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ctgan_fail(CTGAN_E_LAUNCH, "conv16x3p: cannot reserve %zu B of LDS", lds);
        have = lds;
    }
    P16 q = p;
    q.ph_tiles_m = p.M / (TN * 32);
    q.ksplit = 1; q.slab = nullptr;
    hipLaunchKernelGGL((conv16x3p_kernel<TN>), dim3((unsigned)(q.ph_tiles_m * (p.Ng / 128))), dim3(512), lds, st, q, pg);
    ctgan_set_last_kernel(TN == 2 ? "conv16x3p<4x64x128,k32>" : "conv16x3p<4x32x128,k32>");
    ctgan_set_last_symbol("conv16x3p_kernel<%d>", TN);
    return ctgan_check_launch("conv16x3p");
}
int launch_conv16x3p(const P16& p, hipStream_t st) {
    const int bmp = conv16x3p_tile(p);
    PatchGeom pg;
    if (!bmp || !conv16x3p_ok(p, &pg, bmp)) return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv16x3p: shape outside the stride-2 halo form");
    return bmp == 64 ? launch_conv16x3p_t<2>(p, pg, st) : launch_conv16x3p_t<1>(p, pg, st);
}

