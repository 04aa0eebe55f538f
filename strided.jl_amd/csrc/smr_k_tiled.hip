// smr_k_tiled.hip -- family TILED: fused N-ary map whose operands have DIFFERENT unit-stride
// axes (permutedims!, adjoint!, B .= (A .+ A')./2, the 4-way permuted sum).
//
// MI355X design (not the reference's L1-blocked loop nest, src/mapreduce.jl:385-401):
//   * a workgroup owns one N-d tile whose extents are powers of two and which is long enough
//     along EVERY operand's unit-stride axis;
//   * phase 1: each transposed input is read from HBM in ITS OWN stride order (consecutive
//     lanes walk that input's unit-stride axis -> coalesced) and scattered into an LDS tile
//     laid out in DESTINATION order, XOR-swizzled so that the strided LDS writes of a lane
//     group fall on distinct banks;
//   * phase 2: the tile is walked in destination order: staged inputs come from LDS
//     (conflict-free linear reads), inputs that already share the destination's unit axis
//     come straight from HBM, f is applied in registers, the store is coalesced.
// Because the extents are powers of two the element index inside a tile is pure bit slicing,
// and e = r*T + tid splits into a per-thread part (computed once) and a wave-uniform part
// (scalar registers) that are combined with one add (global offset) / one xor (LDS index).
#include "smr_dispatch.h"

#ifndef SMR_CT
#error "compile with -DSMR_CT=0..3"
#endif

namespace smr {

constexpr int MAXT = 5;

struct TiledArgs {
    OpTab ops;
    int32_t N, M, nt, tilelog, thrlog, nstaged, swz, xcd;
    i64 nblocks;
    i64 dims[MAXN];
    i64 ntiles[MAXN];
    int32_t tlogdim[MAXN];
    i64 strides[MAXM][MAXN];
    int32_t tdim[MAXT], tlog[MAXT], lsh[MAXT];
    int32_t staged[MAXM];
    int32_t esh[MAXM][MAXT];  // bit position of tiled dim j inside the enumeration index
                              // of operand k (k = 0: destination order == lsh)
};

SMR_DEV uint32_t lds_swizzle(uint32_t l, int w) {
    if (w == 0) return l;
    const uint32_t x = l >> w;
    const uint32_t f = (x ^ (x >> w) ^ (x >> (2 * w)) ^ (x >> (3 * w))) & ((1u << w) - 1u);
    return l ^ f;
}

template <class T, class F, bool MIXED>
__global__ void __launch_bounds__(256) k_tiled_map(TiledArgs a, F f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);
    const int nin = (F::NIN >= 0) ? F::NIN : a.M - 1;
    const uint32_t tid = threadIdx.x;

    // ---- which tile (XCD-aware: blocks b, b+8, b+16.. share an XCD and hence an L2; give
    // each XCD a contiguous range of tiles so neighbouring tiles share cache lines there)
    i64 b = blockIdx.x;
    if (a.xcd) {
        const i64 per = a.nblocks >> 3;
        b = (b & 7) * per + (b >> 3);
    }
    i64 org[MAXN];  // tile origin per canonical dim (wave-uniform)
    uint32_t lim[MAXT];
#pragma unroll
    for (int j = 0; j < MAXT; ++j) lim[j] = 1;
    bool full = true;
    {
        i64 rem = b;
#pragma unroll
        for (int d = 0; d < MAXN; ++d) {
            org[d] = 0;
            if (d < a.N) {
                const i64 q = rem / a.ntiles[d];
                const i64 t = rem - q * a.ntiles[d];
                rem = q;
                org[d] = t << a.tlogdim[d];
                const i64 left = a.dims[d] - org[d];
                if (left < ((i64)1 << a.tlogdim[d])) full = false;
#pragma unroll
                for (int j = 0; j < MAXT; ++j)
                    if (j < a.nt && a.tdim[j] == d) lim[j] = (uint32_t)(left < 0x7fffffff ? left : 0x7fffffff);
            }
        }
    }
    const int nrep = 1 << (a.tilelog - a.thrlog);

    // ---- phase 1: stage transposed inputs into LDS in destination order ------------------------
#pragma unroll 1
    for (int k = 1; k < MAXM; ++k) {
        if (k >= a.M || a.staged[k] < 0) continue;
        T* L = lds + ((size_t)a.staged[k] << a.tilelog);
        uint32_t ct[MAXT];
        i64 gt = 0;
#pragma unroll
        for (int d = 0; d < MAXN; ++d)
            if (d < a.N) gt += org[d] * a.strides[k][d];
        uint32_t lt = 0;
#pragma unroll
        for (int j = 0; j < MAXT; ++j) {
            ct[j] = 0;
            if (j < a.nt) {
                ct[j] = (tid >> a.esh[k][j]) & ((1u << a.tlog[j]) - 1u);
                gt += (i64)ct[j] * a.strides[k][a.tdim[j]];
                lt |= ct[j] << a.lsh[j];
            }
        }
        lt = lds_swizzle(lt, a.swz);
#pragma unroll 4
        for (int r = 0; r < nrep; ++r) {
            const uint32_t er = (uint32_t)r << a.thrlog;  // wave-uniform
            i64 gr = 0;
            uint32_t lr = 0;
            bool ok = true;
#pragma unroll
            for (int j = 0; j < MAXT; ++j) {
                if (j < a.nt) {
                    const uint32_t cr = (er >> a.esh[k][j]) & ((1u << a.tlog[j]) - 1u);
                    gr += (i64)cr * a.strides[k][a.tdim[j]];
                    lr |= cr << a.lsh[j];
                    ok = ok && ((ct[j] | cr) < lim[j]);
                }
            }
            lr = lds_swizzle(lr, a.swz);
            if (full || ok) L[lt ^ lr] = load_op<T, MIXED>(a.ops, k, gt + gr);
        }
    }
    __syncthreads();

    // ---- phase 2: destination order ---------------------------------------------------------------
    uint32_t ct[MAXT];
    i64 gt[MAXM];
    uint32_t lt = 0;
#pragma unroll
    for (int k = 0; k < MAXM; ++k) {
        gt[k] = 0;
        if (k < a.M && (k == 0 || a.staged[k] < 0)) {
#pragma unroll
            for (int d = 0; d < MAXN; ++d)
                if (d < a.N) gt[k] += org[d] * a.strides[k][d];
        }
    }
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
        ct[j] = 0;
        if (j < a.nt) {
            ct[j] = (tid >> a.lsh[j]) & ((1u << a.tlog[j]) - 1u);
            lt |= ct[j] << a.lsh[j];
#pragma unroll
            for (int k = 0; k < MAXM; ++k)
                if (k < a.M && (k == 0 || a.staged[k] < 0)) gt[k] += (i64)ct[j] * a.strides[k][a.tdim[j]];
        }
    }
    lt = lds_swizzle(lt, a.swz);
#pragma unroll 4
    for (int r = 0; r < nrep; ++r) {
        const uint32_t er = (uint32_t)r << a.thrlog;
        uint32_t lr = 0;
        bool ok = true;
        i64 gr[MAXM];
#pragma unroll
        for (int k = 0; k < MAXM; ++k) gr[k] = 0;
#pragma unroll
        for (int j = 0; j < MAXT; ++j) {
            if (j < a.nt) {
                const uint32_t cr = (er >> a.lsh[j]) & ((1u << a.tlog[j]) - 1u);
                lr |= cr << a.lsh[j];
                ok = ok && ((ct[j] | cr) < lim[j]);
#pragma unroll
                for (int k = 0; k < MAXM; ++k)
                    if (k < a.M && (k == 0 || a.staged[k] < 0)) gr[k] += (i64)cr * a.strides[k][a.tdim[j]];
            }
        }
        lr = lds_swizzle(lr, a.swz);
        if (full || ok) {
            const uint32_t l = lt ^ lr;
            T in[MAXIN];
#pragma unroll
            for (int k = 0; k < MAXIN; ++k) {
                in[k] = T{};
                if (k < nin) {
                    if (a.staged[k + 1] >= 0)
                        in[k] = lds[((size_t)a.staged[k + 1] << a.tilelog) + l];
                    else
                        in[k] = load_op<T, MIXED>(a.ops, k + 1, gt[k + 1] + gr[k + 1]);
                }
            }
            store_op<T, MIXED>(a.ops, gt[0] + gr[0], f(in));
        }
    }
}

template <class T, class F, bool MIXED>
static int go(const Plan& plan, void* const* bases, hipStream_t s, F f) {
    const Canon& c = plan.c;
    const TilePlan& t = plan.tile;
    TiledArgs a;
    std::memset(&a, 0, sizeof a);
    a.ops = make_optab(c, bases);
    a.N = c.N;
    a.M = c.M;
    a.nt = t.nt;
    a.tilelog = t.tilelog;
    int thrlog = 0;
    while ((1 << thrlog) < t.threads) ++thrlog;
    a.thrlog = thrlog;
    a.nstaged = t.nstaged;
    // swizzle width: the 128 B an LDS write group spans, in elements
    {
        int w = 0;
        while ((sizeof(T) << w) < 128) ++w;
        a.swz = (t.tilelog > w) ? w : 0;
    }
    a.nblocks = t.grid;
    a.xcd = (options().xcd_swizzle && (t.grid % 8 == 0) && t.grid >= 16) ? 1 : 0;
    for (int i = 0; i < MAXN; ++i) {
        a.dims[i] = (i < c.N) ? c.dims[i] : 1;
        a.ntiles[i] = (i < c.N) ? t.ntiles[i] : 1;
        a.tlogdim[i] = 0;
    }
    for (int k = 0; k < MAXM; ++k) {
        a.staged[k] = (k < c.M) ? t.staged[k] : -1;
        for (int i = 0; i < MAXN; ++i) a.strides[k][i] = (k < c.M && i < c.N) ? c.strides[k][i] : 0;
    }
    int sh = 0;
    for (int j = 0; j < t.nt; ++j) {
        a.tdim[j] = t.tdim[j];
        a.tlog[j] = t.tlog[j];
        a.lsh[j] = sh;
        a.tlogdim[t.tdim[j]] = t.tlog[j];
        sh += t.tlog[j];
    }
    for (int k = 0; k < c.M; ++k) {
        int pos = 0;
        for (int jj = 0; jj < t.nt; ++jj) {
            int j = t.order[k][jj];
            a.esh[k][j] = pos;
            pos += t.tlog[j];
        }
    }
    size_t lds = (size_t)t.nstaged * ((size_t)1 << t.tilelog) * sizeof(T);
    auto kern = k_tiled_map<T, F, MIXED>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_error(e, "hipFuncSetAttribute(lds)");
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)t.grid), dim3((unsigned)t.threads), lds, s, a, f);
    return check_launch("k_tiled_map");
}

template <>
int launch_tiled_map_ct<SMR_CT>(const Plan& plan, void* const* bases, hipStream_t s) {
    typedef ct_type<SMR_CT>::type T;
    const Canon& c = plan.c;
    if (c.bitcopy) {
#if SMR_CT == SMR_F32
        switch (c.esize[0]) {
            case 1: return go<b8, FIdent<b8>, false>(plan, bases, s, FIdent<b8>{});
            case 2: return go<b16, FIdent<b16>, false>(plan, bases, s, FIdent<b16>{});
            case 4: return go<float, FIdent<float>, false>(plan, bases, s, FIdent<float>{});
            case 8: return go<double, FIdent<double>, false>(plan, bases, s, FIdent<double>{});
            default: return go<c64, FIdent<c64>, false>(plan, bases, s, FIdent<c64>{});
        }
#else
        return set_error(SMR_EINVAL, "bitcopy is dispatched through the f32 object");
#endif
    }
    if (c.mixed) return go<T, FProg<T>, true>(plan, bases, s, FProg<T>{c.prog});
    const unsigned mask = fbit(FK_IDENT) | fbit(FK_ADD2) | fbit(FK_ADD3) | fbit(FK_ADD4) | fbit(FK_SCALE) | fbit(FK_SYM) |
                          fbit(FK_AXPY) | fbit(FK_AXPBY);
    return with_functor<T>(c, mask, [&](auto f) { return go<T, decltype(f), false>(plan, bases, s, f); });
}

}  // namespace smr
