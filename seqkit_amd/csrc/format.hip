// format.hip -- batch -> FASTA/FASTQ output text on the GPU (gfx950).
//
// The step AFTER the per-record hot path: what `seqkit seq` prints for every record
// (seqkit/cmd/seq.go:163-276): the head line, the sequence re-wrapped to the line width
// (byteutil.WrapByteSlice: '\n' after every `width` bytes, none after the last chunk, nothing
// for width < 1 or an empty sequence), and for FASTQ the "+" line and the quality string.
// Three streaming passes: output size of every record, exclusive scan of the sizes, and a
// cooperative copy (16 lanes per record, 16 bytes per lane and step) of names, bases and
// qualities to their places in the output text.  Records longer than SKH_LONG_T leave their
// bases and qualities to a second kernel whose workgroups own windows of the sequence plane, so
// that a chromosome-length record is written by thousands of workgroups, not by 16 lanes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_util.h"
#include "kernels.h"

namespace skh {

namespace {

constexpr int FB = 256;
constexpr int FSCAN_ITEMS = 8;
constexpr uint32_t FSCAN_TILE = FB * FSCAN_ITEMS;

__device__ __forceinline__ uint64_t wrapped_len(uint64_t len, uint32_t width)
{
    return (width < 1 || len == 0) ? len : len + (len - 1) / width;
}

// ID = head line up to the first white space; the whole line if it starts with white space
// (the default --id-regexp `^([^\s]+)\s?` does not match then).  \s is Go's (RE2's) [\t\n\f\r ]: a
// vertical tab is part of the ID.
__device__ __forceinline__ uint32_t id_length(const uint8_t* __restrict__ name, uint32_t n)
{
    uint32_t i = 0;
    while (i < n && !(name[i] == ' ' || (name[i] >= '\t' && name[i] <= '\r' && name[i] != '\v'))) i++;
    return i == 0 ? n : i;
}

__global__ __launch_bounds__(FB) void k_fmt_sizes(const FormatSeqArgs a, uint64_t* __restrict__ size,
                                                  uint32_t* __restrict__ shown_name, uint32_t* __restrict__ n_long)
{
    for (uint64_t r = (uint64_t)blockIdx.x * FB + threadIdx.x; r < a.nrec; r += (uint64_t)gridDim.x * FB) {
        const uint64_t len = a.offsets[r + 1] - a.offsets[r];
        const uint64_t flen = a.filter_offsets ? a.filter_offsets[r + 1] - a.filter_offsets[r] : len;
        const bool keep = !(a.min_len >= 0 && flen < (uint64_t)a.min_len) && !(a.max_len >= 0 && flen > (uint64_t)a.max_len);
        if (keep && len > SKH_LONG_T && (a.print_seq || a.print_qual)) atomicAdd(n_long, 1u);
        uint32_t nl = 0;
        uint64_t sz = 0;
        if (keep) {
            if (a.print_name) {
                nl = a.name_len[r];
                if (a.only_id) nl = id_length(a.text + a.name_pos[r], nl);
                sz += (a.print_seq ? 1 : 0) + (uint64_t)nl + 1;
            }
            if (a.print_seq) sz += wrapped_len(len, a.width) + 1;
            // (no quality plane: an empty quality string, as for a FASTA record printed in FASTQ layout)
            if (a.print_qual && !(a.empty_fasta && len == 0)) sz += (a.plus_line ? 2 : 0) + wrapped_len(a.qual ? len : 0, a.width) + 1;
        }
        size[r] = sz;
        shown_name[r] = nl;
    }
}

// ---- exclusive scan of uint64 (block partials, one-workgroup scan of partials, apply) -------

// the FSCAN_ITEMS consecutive values of a thread: 16-byte loads when all of them exist
__device__ __forceinline__ void scan_items64(const uint64_t* __restrict__ v, uint64_t n, uint64_t base, uint64_t* x)
{
    static_assert(FSCAN_ITEMS % 2 == 0, "pairs of values per load");
    if (base + FSCAN_ITEMS <= n) {
#pragma unroll
        for (int k = 0; k < FSCAN_ITEMS; k += 2) {
            const uint4 t = ldu16(reinterpret_cast<const uint8_t*>(v + base + k));     // (callers' arrays: 8-byte aligned)
            x[k] = ((uint64_t)t.y << 32) | t.x;
            x[k + 1] = ((uint64_t)t.w << 32) | t.z;
        }
    } else {
#pragma unroll
        for (int k = 0; k < FSCAN_ITEMS; k++) x[k] = base + k < n ? v[base + k] : 0;
    }
}

__global__ __launch_bounds__(FB) void k_scan_partials64(const uint64_t* __restrict__ v, uint64_t n,
                                                        uint64_t* __restrict__ part)
{
    __shared__ uint64_t s_w[FB / 64];
    const uint64_t base = (uint64_t)blockIdx.x * FSCAN_TILE + threadIdx.x * FSCAN_ITEMS;
    uint64_t x[FSCAN_ITEMS], s = 0;
    scan_items64(v, n, base, x);
#pragma unroll
    for (int k = 0; k < FSCAN_ITEMS; k++) s += x[k];
    for (int d = 1; d < 64; d <<= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < FB / 64; w++) t += s_w[w];
        part[blockIdx.x] = t;
    }
}

// Exclusive scan of the workgroup sums, in place, by one workgroup: rounds of 1024 consecutive entries (one per
// thread, the next round's entry loaded before this round is scanned), a shuffle scan per wavefront, the 16
// wavefront totals through LDS, the running total carried in registers.
__global__ __launch_bounds__(1024) void k_scan_top64(uint64_t* __restrict__ part, uint64_t npart, uint64_t* __restrict__ total)
{
    __shared__ uint64_t s_w[16];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t carry = 0;
    uint64_t nxt = threadIdx.x < npart ? part[threadIdx.x] : 0;
    for (uint64_t base = 0; base < npart; base += 1024) {
        const uint64_t i = base + threadIdx.x;
        const uint64_t v = nxt;
        nxt = i + 1024 < npart ? part[i + 1024] : 0;
        uint64_t inc = v;
        for (int d = 1; d < 64; d <<= 1) { const uint64_t t = __shfl_up(inc, d); if ((int)lane >= d) inc += t; }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        uint64_t run = carry + inc - v;
#pragma unroll
        for (uint32_t w = 0; w < 16; w++) {
            const uint64_t t = s_w[w];
            if (w < wave) run += t;
            carry += t;
        }
        if (i < npart) part[i] = run;
        __syncthreads();                                  // the totals are rewritten in the next round
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(FB) void k_scan_apply64(const uint64_t* __restrict__ v, uint64_t n,
                                                     const uint64_t* __restrict__ part, uint64_t* __restrict__ out)
{
    __shared__ uint64_t s_w[FB / 64];
    const uint64_t base = (uint64_t)blockIdx.x * FSCAN_TILE + threadIdx.x * FSCAN_ITEMS;
    uint64_t x[FSCAN_ITEMS], s = 0;
    scan_items64(v, n, base, x);
#pragma unroll
    for (int k = 0; k < FSCAN_ITEMS; k++) s += x[k];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t inc = s;
    for (int d = 1; d < 64; d <<= 1) { const uint64_t t = __shfl_up(inc, d); if ((int)lane >= d) inc += t; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint64_t run = part[blockIdx.x] + inc - s;
    for (uint32_t w = 0; w < wave; w++) run += s_w[w];
    if (base + FSCAN_ITEMS <= n) {                         // 16-byte stores
#pragma unroll
        for (int k = 0; k < FSCAN_ITEMS; k += 2) {
            const uint64_t r0 = run, r1 = run + x[k];
            const uint4 t = make_uint4((uint32_t)r0, (uint32_t)(r0 >> 32), (uint32_t)r1, (uint32_t)(r1 >> 32));
            __builtin_memcpy(out + base + k, &t, 16);
            run = r1 + x[k + 1];
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < FSCAN_ITEMS; k++) { if (base + k < n) out[base + k] = run; run += x[k]; }
}

// ---- cooperative writers (G lanes of one record) --------------------------------------------

__device__ __forceinline__ void copy16(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src)
{
    uint4 v;
    __builtin_memcpy(&v, src, 16);
    __builtin_memcpy(dst, &v, 16);
}

// Tables of a record, read by the first lane of its group and handed to the others by cross-lane
// reads: the L1 counts every lane of a load, also when all of them ask for the same address, and
// these kernels run at its access rate.
__device__ __forceinline__ uint64_t from_lane64(uint64_t v, uint32_t src_lane)
{
    return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)src_lane) << 32) |
           (uint32_t)__shfl((int)(uint32_t)v, (int)src_lane);
}

// p / width for the positions inside one record, for the records whose bases the lane-group and LDS-staged
// writers copy (at most SKH_LONG_T = 2^15 bytes), where one multiplication by
// ceil(2^32 / width) is exact (p * (magic * width - 2^32) < 2^32 holds for width <= 2^17); a width beyond that
// never divides such a position but to 0.
struct Div32 {
    uint32_t width, magic;
    __device__ __forceinline__ explicit Div32(uint32_t w)
        : width(w), magic(w < 2 || w > 131072 ? 0u : (uint32_t)((0x100000000ull + w - 1) / w)) {}
    __device__ __forceinline__ uint32_t operator()(uint32_t p) const { return width == 1 ? p : __umulhi(p, magic); }
};

template <typename T>
__device__ __forceinline__ T wrapped_len_t(T len, uint32_t width, T breaks)
{
    return (width < 1 || len == 0) ? len : len + breaks;
}

__device__ __forceinline__ uint32_t wave_max32(uint32_t v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)v, d); v = t > v ? t : v; }
    return v;
}

// One 16-byte piece (bytes [p, p + 16) of a field of L bytes, p < L) to its place: plain, or with a '\n' after
// every `width` bytes but the last line (the piece that holds the base before a break writes the break).
// The end of a field goes as its last 16 bytes once more (same values); what fits neither way byte by byte.
// In two halves, so that a caller can issue the loads of several pieces before the first store: the plan
// (a pure function of L, p and the width), the loads it asks for, and the stores.
struct PiecePlan {
    uint32_t n16;                    // 16-byte copies: 0, 1 or 2
    uint32_t so0, do0, so1, do1;     // their offsets in the field and in its place
    uint32_t last, l0;               // last byte of the piece, line of its first byte
    bool bytewise;                   // bytes [p, last] one by one
};

__device__ __forceinline__ PiecePlan plan_piece(uint32_t L, uint32_t p, bool wrapped, const Div32& div)
{
    PiecePlan pl{0, 0, 0, 0, 0, 0, 0, false};
    pl.last = p + 16 <= L ? p + 15 : L - 1;
    if (!wrapped) {
        if (p + 16 <= L) { pl.n16 = 1; pl.so0 = pl.do0 = p; }
        else if (L >= 16) { pl.n16 = 1; pl.so0 = pl.do0 = L - 16; }
        else pl.bytewise = true;
        return pl;
    }
    const uint32_t width = div.width;
    const uint32_t l0 = div(p), l1 = div(pl.last);
    pl.l0 = l0;
    if (l0 == l1 && p + 16 <= L) {
        pl.n16 = 1; pl.so0 = p; pl.do0 = p + l0;
    } else if (l0 == l1 && L >= 16 && div(L - 16) == l0) {            // the end of the field, within one line
        pl.n16 = 1; pl.so0 = L - 16; pl.do0 = L - 16 + l0;
    } else if (l1 == l0 + 1 && width >= 16 && (l0 + 1) * width >= 16 && (l0 + 1) * width + 16 <= L) {
        const uint32_t b = (l0 + 1) * width;          // first base of the next line: the 16 bytes that end the line and
        pl.n16 = 2;                                   // the 16 that start the next one (overlaps carry the same values)
        pl.so0 = b - 16; pl.do0 = b - 16 + l0;
        pl.so1 = b; pl.do1 = b + l1;
    } else {
        pl.bytewise = true;
    }
    return pl;
}

__device__ __forceinline__ void store_piece(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t L, uint32_t p,
                                            bool wrapped, const Div32& div, const PiecePlan& pl, uint4 v0, uint4 v1)
{
    if (pl.n16 >= 1) __builtin_memcpy(dst + pl.do0, &v0, 16);
    if (pl.n16 == 2) __builtin_memcpy(dst + pl.do1, &v1, 16);
    if (!wrapped) {
        if (pl.bytewise) for (uint32_t k = p; k < L; k++) dst[k] = src[k];
        return;
    }
    if (pl.bytewise) for (uint32_t k = p; k <= pl.last; k++) dst[k + div(k)] = src[k];
    const uint32_t width = div.width, breaks = div(L - 1);
    for (uint32_t b = pl.l0 + 1; b <= breaks && b * width - 1 <= pl.last; b++) dst[b * width + b - 1] = '\n';
}

__device__ __forceinline__ void copy_piece(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t L, uint32_t p,
                                           bool wrapped, const Div32& div)
{
    const PiecePlan pl = plan_piece(L, p, wrapped, div);
    const uint4 v0 = pl.n16 >= 1 ? ldu16(src + pl.so0) : make_uint4(0, 0, 0, 0);
    const uint4 v1 = pl.n16 == 2 ? ldu16(src + pl.so1) : make_uint4(0, 0, 0, 0);
    store_piece(dst, src, L, p, wrapped, div, pl, v0, v1);
}

// One field (names, bases or qualities) of the n records of a step copied into the stage, its 16-byte pieces
// dealt over the 64 lanes: the field gets PF piece slots per record (the most any record of the step needs),
// slot q = lane + 64 k is piece q mod PF of record q / PF.  field_of(r, L, at, src), called by all lanes
// together (it reads across lanes), gives the field of record r: its length, its place in the stage and its
// source.  PU rounds of slots at a time, the loads of all of them ahead of the first write to the stage.
// A field of n <= SKH_LONG_T bytes straight to global memory by the G lanes of a record's group (lane j): piece
// j + G k each, the loads of four pieces ahead of the first store; the pieces write their own line breaks.
template <uint32_t G>
__device__ __forceinline__ void copy_field_direct(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t n,
                                                  bool wrapped, const Div32& div, uint32_t j)
{
    static_assert(SKH_LONG_T <= 32768, "Div32 is exact up to 2^15");
    constexpr int PU = 4;
    for (uint32_t p0 = j * 16; p0 < n; p0 += G * 16 * PU) {
        uint4 v0[PU], v1[PU];
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const uint32_t p = p0 + u * G * 16;
            v0[u] = v1[u] = make_uint4(0, 0, 0, 0);
            if (p < n) {
                const PiecePlan pl = plan_piece(n, p, wrapped, div);
                if (pl.n16 >= 1) v0[u] = ldu16(src + pl.so0);
                if (pl.n16 == 2) v1[u] = ldu16(src + pl.so1);
            }
        }
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const uint32_t p = p0 + u * G * 16;
            if (p < n) store_piece(dst, src, n, p, wrapped, div, plan_piece(n, p, wrapped, div), v0[u], v1[u]);
        }
    }
}

// One record's text at o (where its first byte goes), by the G lanes of its group; lane j of the group.
// Bases and qualities of a record longer than SKH_LONG_T are left to k_fmt_write_long.
template <uint32_t G>
__device__ __forceinline__ void write_record_text(const FormatSeqArgs& a, uint8_t* __restrict__ o, uint64_t s, uint64_t len,
                                                  uint32_t nl, uint64_t name, const Div32& div, uint32_t j)
{
    if (a.print_name) {
        if (a.print_seq) { if (j == 0) *o = (a.empty_fasta && len == 0) ? (uint8_t)'>' : a.mark; o++; }
        copy_field_direct<G>(o, a.text + name, nl, false, div, j);
        if (j == 0) o[nl] = '\n';
        o += nl + 1;
    }
    const bool is_long = len > SKH_LONG_T;
    const bool wrapped = a.width >= 1;
    const uint64_t w = wrapped_len_t<uint64_t>(len, a.width, (!wrapped || len == 0) ? 0 : (is_long ? (len - 1) / a.width : div((uint32_t)len - 1)));
    if (a.print_seq) {
        if (!is_long) copy_field_direct<G>(o, a.seq + s, (uint32_t)len, wrapped, div, j);
        if (j == 0) o[w] = '\n';
        o += w + 1;
    }
    if (a.print_qual && !(a.empty_fasta && len == 0)) {
        if (a.plus_line) { if (j == 0) { o[0] = '+'; o[1] = '\n'; } o += 2; }
        if (a.qual) {
            if (!is_long) copy_field_direct<G>(o, a.qual + s, (uint32_t)len, wrapped, div, j);
            if (j == 0) o[w] = '\n';
        } else if (j == 0) {
            o[0] = '\n';                      // no quality plane: an empty quality string
        }
    }
}

constexpr int FMT_ROUNDS = 1;        // slot rounds per batch in the two formatters (see k_fmt_write)

template <int PU, typename F>
__device__ __forceinline__ void flat_copy_field(uint8_t* __restrict__ stage, uint32_t lane, uint32_t n, uint32_t PF, bool wrap,
                                                const Div32& div, F&& field_of)
{
    const Div32 by_pf(PF);
    const uint32_t nslots = n * PF;
    for (uint32_t q0 = 0; q0 < nslots; q0 += 64 * PU) {
        uint4 v0[PU], v1[PU];
        const uint8_t* src[PU];
        uint32_t at[PU], L[PU], P[PU];
        bool act[PU];
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const uint32_t qq = q0 + 64 * u + lane;
            const uint32_t q = qq < nslots ? qq : nslots - 1;
            const uint32_t r = by_pf(q), k = q - r * PF;
            field_of(r, L[u], at[u], src[u]);
            P[u] = 16 * k;
            act[u] = qq < nslots && P[u] < L[u];
            v0[u] = v1[u] = make_uint4(0, 0, 0, 0);
            if (act[u]) {
                const PiecePlan pl = plan_piece(L[u], P[u], wrap, div);
                if (pl.n16 >= 1) v0[u] = ldu16(src[u] + pl.so0);
                if (pl.n16 == 2) v1[u] = ldu16(src[u] + pl.so1);
            }
        }
#pragma unroll
        for (int u = 0; u < PU; u++) {
            if (!act[u]) continue;
            const PiecePlan pl = plan_piece(L[u], P[u], wrap, div);
            store_piece(stage + at[u], src[u], L[u], P[u], wrap, div, pl, v0[u], v1[u]);
        }
    }
}

// The text of the n records of a step (lane i holds record i: op = where its text starts in the stage, sz = its
// size, 0 for a record that is not printed) put together in the stage.  The separators are written by the
// record's own lane; the 16-byte pieces of names, bases and qualities of ALL records are dealt over the 64
// lanes, field by field: a field gets PF piece slots per record (the most any record of the step needs), slot
// q = lane + 64 k is piece q mod PF of record q / PF -- lanes are busy whatever a field's length, where 16
// lanes per record keep about half their slots filled (a name is 3 pieces, a read 10).
__device__ __forceinline__ void write_step_flat(const FormatSeqArgs& a, uint8_t* __restrict__ stage, uint32_t lane, uint32_t n,
                                                const Div32& div, uint32_t op, uint32_t sz, uint32_t len, uint32_t nl,
                                                uint64_t s, uint64_t name)
{
    const bool has = lane < n && sz != 0;
    if (!has) { len = 0; nl = 0; }
    const bool wrapped = a.width >= 1;
    const bool with_qual = a.print_qual && a.qual != nullptr;
    const uint32_t PN = wave_max32(a.print_name ? (nl + 15) >> 4 : 0u);
    const uint32_t PS = wave_max32(a.print_seq ? (len + 15) >> 4 : 0u);
    const uint32_t PQ = with_qual ? wave_max32((len + 15) >> 4) : 0u;
    // where a record's fields start, from its name and sequence lengths
    auto layout = [&](uint32_t nl_r, uint32_t len_r, uint32_t& o_name, uint32_t& o_seq, uint32_t& o_plus, uint32_t& o_qual,
                      uint32_t& w) {
        o_name = (a.print_name && a.print_seq) ? 1u : 0u;
        o_seq = a.print_name ? o_name + nl_r + 1 : 0u;
        w = (!wrapped || len_r == 0) ? len_r : len_r + div(len_r - 1);
        o_plus = o_seq + (a.print_seq ? w + 1 : 0u);
        o_qual = o_plus + (a.plus_line ? 2u : 0u);
    };
    if (has) {
        uint32_t o_name, o_seq, o_plus, o_qual, w;
        layout(nl, len, o_name, o_seq, o_plus, o_qual, w);
        uint8_t* o = stage + op;
        if (a.print_name) {
            if (a.print_seq) o[0] = (a.empty_fasta && len == 0) ? (uint8_t)'>' : a.mark;
            o[o_name + nl] = '\n';
        }
        if (a.print_seq) o[o_seq + w] = '\n';
        if (a.print_qual && !(a.empty_fasta && len == 0)) {
            if (a.plus_line) { o[o_plus] = '+'; o[o_plus + 1] = '\n'; }
            o[o_qual + (a.qual ? w : 0u)] = '\n';
        }
    }
    const uint64_t s0 = from_lane64(s, 0);               // sequences of a step lie within 4 GiB of each other
    const uint32_t s_rel = (uint32_t)(s - s0);
    // field by field, so that the lanes of one instruction read one kind of source
#pragma unroll 1
    for (int f = 0; f < 3; f++) {
        const uint32_t PF = f == 0 ? PN : (f == 1 ? PS : PQ);
        if (PF == 0) continue;
        flat_copy_field<FMT_ROUNDS>(stage, lane, n, PF, f != 0 && wrapped, div,
                           [&](uint32_t r, uint32_t& L, uint32_t& at, const uint8_t*& src) {
            const uint32_t op_r = (uint32_t)__shfl((int)op, (int)r), len_r = (uint32_t)__shfl((int)len, (int)r);
            const uint32_t nl_r = (uint32_t)__shfl((int)nl, (int)r), srel_r = (uint32_t)__shfl((int)s_rel, (int)r);
            const uint64_t name_r = f == 0 ? from_lane64(name, r) : 0;
            uint32_t o_name, o_seq, o_plus, o_qual, w;
            layout(nl_r, len_r, o_name, o_seq, o_plus, o_qual, w);
            L = f == 0 ? nl_r : len_r;
            at = op_r + (f == 0 ? o_name : (f == 1 ? o_seq : o_qual));
            src = f == 0 ? a.text + name_r : (f == 1 ? a.seq : a.qual) + s0 + srel_r;
        });
    }
}

// The output text of consecutive records is one contiguous stretch.  A wavefront takes as many records
// per step as fit its LDS stage (from the mean record size, at most 64), puts their text together there
// -- the pieces of a record start at any byte, which costs LDS next to nothing, while every misaligned
// or partial store to global memory is a request of its own to the memory pipeline -- and writes the
// stretch out as aligned 16-byte chunks in whole-wavefront rows; the bytes at either end that share a
// chunk with the neighbouring step go byte by byte.  A step whose text does not fit the stage (long
// records) is counted and left to k_fmt_write_direct.
constexpr uint32_t FW_STAGE = 7 * 1024;

__device__ __forceinline__ uint32_t fmt_records_per_step(uint64_t total, uint64_t nrec)
{
    const uint64_t mean = nrec ? total / nrec + 1 : 1;
    const uint64_t r = (FW_STAGE - 64) * 7 / 8 / mean;
    return r < 1 ? 1u : (r > 64 ? 64u : (uint32_t)r);
}

// Records (rows) far beyond the stage on average: the staged writers leave at once and the direct kernels
// take every record.
__device__ __forceinline__ bool all_steps_direct(uint64_t total, uint64_t nrec) { return nrec != 0 && total / nrec > 4 * FW_STAGE; }

// true when the writer that puts rows together in LDS left the step of record r to the direct kernels
__device__ __forceinline__ bool step_is_direct(uint64_t r, uint32_t R, uint64_t nrec, const uint64_t* __restrict__ pos,
                                               const uint64_t* __restrict__ size)
{
    const uint64_t r0 = r / R * R, rl = (nrec - r0 < R ? nrec : r0 + R) - 1;
    return pos[rl] + size[rl] - (pos[r0] & ~15ull) > FW_STAGE;
}

// (five wavefronts per SIMD is what the stages of five workgroups leave room for; the register allocation is held
// to that -- 96 registers, nothing spilled -- because this kernel gains more from a fifth resident wavefront than
// from loads issued further ahead: 0.85 -> 0.77 ms on 4 M x 150 bp FASTQ, where FMT_ROUNDS = 2 at four wavefronts
// gives 0.81 ms and at five, with spills, 0.80 ms)
__attribute__((amdgpu_waves_per_eu(5, 5)))
__global__ __launch_bounds__(FB) void k_fmt_write(const FormatSeqArgs a, const uint64_t* __restrict__ size,
                                                  const uint64_t* __restrict__ pos, const uint32_t* __restrict__ shown_name,
                                                  const uint64_t* __restrict__ total, uint32_t* __restrict__ n_direct,
                                                  uint8_t* __restrict__ out, uint64_t out_capacity)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[FB / 64][FW_STAGE + 32];
    const uint64_t tot = *total;
    if (tot > out_capacity) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t R = fmt_records_per_step(tot, a.nrec);
    const uint64_t nsteps = (a.nrec + R - 1) / R;
    if (all_steps_direct(tot, a.nrec)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *n_direct = 1u;
        return;
    }
    const Div32 div(a.width);
    uint8_t* stage = s_stage[wave];
    for (uint64_t step = (uint64_t)blockIdx.x * (FB / 64) + wave; step < nsteps; step += (uint64_t)gridDim.x * (FB / 64)) {
        const uint64_t r0 = step * R;
        const uint32_t n = a.nrec - r0 < R ? (uint32_t)(a.nrec - r0) : R;
        // lane i holds the tables of record r0 + i
        uint64_t t_pos = 0, t_end = 0, t_s = 0, t_name = 0;
        uint32_t t_nl = 0, t_len = 0;
        if (lane < n) {
            const uint64_t r = r0 + lane;
            const uint64_t s = a.offsets[r], e = a.offsets[r + 1];
            t_pos = pos[r]; t_end = t_pos + size[r]; t_s = s;
            t_len = e - s > 0xffffffffull ? 0xffffffffu : (uint32_t)(e - s);      // only its bytes that are printed count here
            if (a.print_name) { t_nl = shown_name[r]; t_name = a.name_pos[r]; }
        }
        const uint64_t p0 = from_lane64(t_pos, 0), p1 = from_lane64(t_end, n - 1);
        const uint64_t a0 = p0 & ~15ull;
        if (p1 - a0 > FW_STAGE) {                          // wave-uniform
            if (lane == 0) *n_direct = 1u;                 // (every writer stores the same value)
            continue;
        }
        const uint32_t t_size = (uint32_t)(t_end - t_pos);
        if (p1 == p0) continue;
        write_step_flat(a, stage, lane, n, div, (uint32_t)(t_pos - a0), t_size, t_len, t_nl, t_s, t_name);
        wave_lds_handover();
        const uint32_t nbytes = (uint32_t)(p1 - a0), head = (uint32_t)(p0 - a0);
        const uint32_t c_lo = head ? 16u : 0u, c_hi = nbytes & ~15u;
        for (uint32_t c = lane * 16; c < c_hi; c += 1024)
            if (c >= c_lo) *reinterpret_cast<uint4*>(out + a0 + c) = *reinterpret_cast<const uint4*>(stage + c);
        if (head && lane >= head && lane < 16 && lane < nbytes) out[a0 + lane] = stage[lane];
        if (c_hi < nbytes && (c_hi != 0 || !head) && lane < nbytes - c_hi) out[a0 + c_hi + lane] = stage[c_hi + lane];
        wave_lds_handover();
    }
}

// The records of the steps k_fmt_write left out (their text does not fit the LDS stage), written straight
// to global memory by 16 lanes per record.
__global__ __launch_bounds__(FB) void k_fmt_write_direct(const FormatSeqArgs a, const uint64_t* __restrict__ size,
                                                         const uint64_t* __restrict__ pos, const uint32_t* __restrict__ shown_name,
                                                         const uint64_t* __restrict__ total, const uint32_t* __restrict__ n_direct,
                                                         uint8_t* __restrict__ out, uint64_t out_capacity)
{
    constexpr uint32_t G = 16;
    const uint64_t tot = *total;
    if (*n_direct == 0 || tot > out_capacity) return;
    const uint32_t j = threadIdx.x & (G - 1);
    const uint32_t leader = (threadIdx.x & 63) & ~(G - 1);
    const uint32_t R = fmt_records_per_step(tot, a.nrec);
    const bool all_direct = all_steps_direct(tot, a.nrec);
    const Div32 div(a.width);
    for (uint64_t r = ((uint64_t)blockIdx.x * FB + threadIdx.x) / G; r < a.nrec; r += (uint64_t)gridDim.x * (FB / G)) {
        uint64_t m_size = 0, m_pos = 0, m_s = 0, m_e = 0, m_name = 0;
        uint32_t m_nl = 0;
        if (j == 0) {
            if (all_direct || step_is_direct(r, R, a.nrec, pos, size)) m_size = size[r];
            if (m_size != 0) {
                m_pos = pos[r]; m_s = a.offsets[r]; m_e = a.offsets[r + 1];
                if (a.print_name) { m_nl = shown_name[r]; m_name = a.name_pos[r]; }
            }
        }
        if (from_lane64(m_size, leader) == 0) continue;
        const uint64_t s = from_lane64(m_s, leader), len = from_lane64(m_e, leader) - s;
        write_record_text<G>(a, out + from_lane64(m_pos, leader), s, len, __shfl(m_nl, leader),
                                              from_lane64(m_name, leader), div, j);
    }
}

// Bases [k0, k1) of a record's sequence (or quality) string `src` to their places in the wrapped
// text at `dst` (where base 0 goes), by the whole workgroup; the line break that follows base
// b*width-1 is written by whoever writes that base.  n = length of the whole string.
__device__ __forceinline__ void copy_wrapped_range(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint64_t n,
                                                   uint32_t width, uint64_t k0, uint64_t k1)
{
    const bool wrap = width >= 1;
    for (uint64_t p = k0 + 16ull * threadIdx.x; p < k1; p += 16ull * FB) {
        const uint64_t last = p + 16 <= k1 ? p + 15 : k1 - 1;
        const uint64_t l0 = wrap ? p / width : 0, l1 = wrap ? last / width : 0;
        if (l0 == l1 && p + 16 <= k1) {
            uint4 v;
            __builtin_memcpy(&v, src + p, 16);
            __builtin_memcpy(dst + p + l0, &v, 16);
        } else {
            for (uint64_t k = p; k <= last; k++) dst[k + (wrap ? k / width : 0)] = src[k];
        }
    }
    if (wrap && n != 0) {
        // break b (1-based) follows base b*width-1 and sits at dst[b*width + b - 1]
        const uint64_t breaks = (n - 1) / width;
        const uint64_t b0 = k0 / width + 1;                 // first b with b*width-1 >= k0
        for (uint64_t b = b0 + threadIdx.x; b <= breaks && b * width - 1 < k1; b += FB) dst[b * width + b - 1] = '\n';
    }
}

// Bases and qualities of the records longer than SKH_LONG_T.  Workgroups own aligned windows of
// the sequence plane; only the records that hold a window's first and last byte can be long.
__global__ __launch_bounds__(FB) void k_fmt_write_long(const FormatSeqArgs a, const uint64_t* __restrict__ size,
                                                       const uint64_t* __restrict__ pos,
                                                       const uint32_t* __restrict__ shown_name,
                                                       const uint64_t* __restrict__ total, const uint32_t* __restrict__ n_long,
                                                       uint8_t* __restrict__ out, uint64_t out_capacity)
{
    if (*n_long == 0 || *total > out_capacity) return;
    const uint64_t plane = a.offsets[a.nrec];
    const uint64_t ntiles = (plane + SKH_TILE - 1) / SKH_TILE;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t t0 = tile * SKH_TILE;
        const uint64_t t1 = t0 + SKH_TILE < plane ? t0 + SKH_TILE : plane;
        const uint64_t r_lo = find_record(a.offsets, a.nrec, t0);
        const uint64_t r_hi = a.offsets[r_lo + 1] > t1 - 1 ? r_lo : find_record(a.offsets, a.nrec, t1 - 1);
#pragma unroll 1
        for (int c = 0; c < 2; c++) {
            if (c == 1 && r_hi == r_lo) break;
            const uint64_t r = c == 0 ? r_lo : r_hi;
            const uint64_t s = a.offsets[r], e = a.offsets[r + 1], len = e - s;
            if (len <= SKH_LONG_T || size[r] == 0) continue;          // block-uniform
            const uint64_t k0 = (s > t0 ? s : t0) - s, k1 = (e < t1 ? e : t1) - s;
            uint8_t* o = out + pos[r];
            if (a.print_name) o += (a.print_seq ? 1 : 0) + (uint64_t)shown_name[r] + 1;
            if (a.print_seq) {
                copy_wrapped_range(o, a.seq + s, len, a.width, k0, k1);
                o += wrapped_len(len, a.width) + 1;
            }
            if (a.print_qual && a.qual) {
                if (a.plus_line) o += 2;
                copy_wrapped_range(o, a.qual + s, len, a.width, k0, k1);
            }
        }
    }
}

// ---- fx2tab rows ------------------------------------------------------------------------------

// number of decimal digits of v, and v in decimal at buf[0 .. nd) (32-bit arithmetic whenever v fits)
__device__ __forceinline__ uint32_t ndigits(uint64_t v)
{
    if (v >> 32) {
        uint32_t n = 1;
        while (v >= 10) { v /= 10; n++; }
        return n;
    }
    const uint32_t x = (uint32_t)v;                       // comparisons, no loop whose length differs between lanes
    return 1u + (x >= 10u) + (x >= 100u) + (x >= 1000u) + (x >= 10000u) + (x >= 100000u) + (x >= 1000000u) +
           (x >= 10000000u) + (x >= 100000000u) + (x >= 1000000000u);
}

__device__ __forceinline__ void put_decimal(uint64_t v, uint32_t nd, uint8_t* buf)
{
    if (v >> 32) {
        for (uint32_t k = nd; k-- > 0;) { buf[k] = (uint8_t)('0' + v % 10); v /= 10; }
        return;
    }
    uint32_t x = (uint32_t)v;
    for (uint32_t k = nd; k-- > 0;) { buf[k] = (uint8_t)('0' + x % 10); x /= 10; }
}

// "%.2f" of v, exactly as C and Go print it: the binary value is rounded to hundredths with
// ties to even on the exact value (integer arithmetic; |v| < 2^52 here: percentages).
struct Fixed2 { uint64_t ip; uint32_t fp; uint32_t kind; bool neg; };     // kind: 0 = number, 1 = NaN, 2 = infinity

__device__ __forceinline__ Fixed2 fixed2_of(double v)
{
    const uint64_t bits = (uint64_t)__double_as_longlong(v);
    Fixed2 f;
    f.neg = (bits >> 63) != 0;
    f.ip = 0; f.fp = 0; f.kind = 0;
    const uint32_t E = (uint32_t)((bits >> 52) & 0x7ff);
    const uint64_t frac = bits & ((1ull << 52) - 1);
    if (E == 0x7ff) { f.kind = frac ? 1u : 2u; return f; }
    // |v| = m * 2^-sh
    const uint64_t m = E ? (frac | (1ull << 52)) : frac;
    const int sh = E ? 1075 - (int)E : 1074;
    uint64_t q = 0;                                   // |v| * 100 rounded to an integer
    const uint64_t P = m * 100;                       // < 2^60
    if (sh <= 0) q = P << (-sh);                      // not reached for percentages (|v| >= 2^52)
    else if (sh < 64) {
        q = P >> sh;
        const uint64_t rem = P & ((1ull << sh) - 1), half = 1ull << (sh - 1);
        if (rem > half || (rem == half && (q & 1))) q++;
    }
    if (q >> 32) { f.ip = q / 100; f.fp = (uint32_t)(q % 100); }
    else { const uint32_t q32 = (uint32_t)q; f.ip = q32 / 100; f.fp = q32 % 100; }
    return f;
}

// "%.2f" of 100 * sum / len without the double: the text the arithmetic above prints, whenever this returns true.
// The reference prints float64(sum) / float64(len) * 100 (for GC content float64(g) / float64(len) +
// float64(c) / float64(len), times 100): two to four roundings of 2^-53 relative each, so the value whose
// hundredths are printed is x = 10000 * sum / len within 5e-16 * x <= 5e-12 (sum <= len).  The printed hundredths
// are x rounded to an integer, and x lies at least 1 / (2 len) > 7e-6 (len < 65536) from the nearest k + 1/2
// unless it IS k + 1/2 -- then the roundings of the doubles decide and this returns false.  So apart from exact
// ties, rounding the exact quotient gives the same digits: one 32-bit division with remainder (estimate from a
// float reciprocal of len, off by one at most, put right by the remainder).
__device__ __forceinline__ bool fixed2_ratio_fast(uint32_t sum, uint32_t len, float rlen, Fixed2& f)
{
    const uint32_t N = sum * 10000u;                      // < 2^32: sum <= len < 65536
    uint32_t q = (uint32_t)((float)N * rlen);             // 10000 at most: the estimate is within 0.01 of N / len
    uint32_t r = N - q * len;
    if ((int32_t)r < 0) { q--; r += len; }
    else if (r >= len) { q++; r -= len; }
    if (2u * r == len) return false;                      // a tie at the hundredths: the doubles decide
    q += 2u * r > len ? 1u : 0u;
    const uint32_t ip = q / 100u;
    f.neg = false; f.kind = 0; f.ip = ip; f.fp = q - 100u * ip;
    return true;
}

// characters of the text of f; with PUT also written to buf (at most 24)
template <bool PUT>
__device__ __forceinline__ uint32_t fixed2_text(const Fixed2& f, uint8_t* buf)
{
    if (f.kind == 1) { if (PUT) { buf[0] = 'N'; buf[1] = 'a'; buf[2] = 'N'; } return 3; }
    if (f.kind == 2) { if (PUT) { buf[0] = f.neg ? '-' : '+'; buf[1] = 'I'; buf[2] = 'n'; buf[3] = 'f'; } return 4; }
    uint32_t n = 0;
    if (f.neg) { if (PUT) buf[0] = '-'; n = 1; }
    uint32_t nd;
    if (f.ip < 1000) {                                    // percentages: three digits at most, no loops
        const uint32_t ip = (uint32_t)f.ip, h = ip / 100, t = ip / 10 - 10 * h, u = ip - 10 * (ip / 10);
        nd = 1u + (ip >= 10u) + (ip >= 100u);
        if (PUT) {
            uint8_t* o = buf + n;
            if (ip >= 100u) *o++ = (uint8_t)('0' + h);
            if (ip >= 10u) *o++ = (uint8_t)('0' + t);
            *o = (uint8_t)('0' + u);
        }
    } else {
        nd = ndigits(f.ip);
        if (PUT) put_decimal(f.ip, nd, buf + n);
    }
    if (PUT) {
        buf[n + nd] = '.';
        buf[n + nd + 1] = (uint8_t)('0' + f.fp / 10);
        buf[n + nd + 2] = (uint8_t)('0' + f.fp % 10);
    }
    return n + nd + 3;
}

// The counts of counting pass 0 for one record, loaded before anything uses them: the columns come one after the
// other, and a load per column, each waited for, is a chain of memory latencies per record -- which is what the
// row kernels spent their time on (a wavefront has 64 rows to do and little else to overlap).  Passes beyond the
// first (more than 8 classes asked for) are read where they are used.
typedef uint64_t CountRow __attribute__((ext_vector_type(8)));       // (a vector, not an array: it stays in registers)

__device__ __forceinline__ CountRow load_count_row(const FormatTabArgs& a, uint64_t r)
{
    CountRow row = 0;
    const uint64_t* p = a.counts[0];
    const uint32_t nc = a.nclass[0];
    if (p == nullptr) return row;
    p += r * nc;
    if (nc == 4 && (reinterpret_cast<uintptr_t>(a.counts[0]) & 15) == 0) {       // (the same for every lane)
        const uint4 lo = *reinterpret_cast<const uint4*>(p), hi = *reinterpret_cast<const uint4*>(p + 2);
        row.s0 = lo.x | ((uint64_t)lo.y << 32); row.s1 = lo.z | ((uint64_t)lo.w << 32);
        row.s2 = hi.x | ((uint64_t)hi.y << 32); row.s3 = hi.z | ((uint64_t)hi.w << 32);
    } else {
        if (nc > 0) row.s0 = p[0];
        if (nc > 1) row.s1 = p[1];
        if (nc > 2) row.s2 = p[2];
        if (nc > 3) row.s3 = p[3];
        if (nc > 4) row.s4 = p[4];
        if (nc > 5) row.s5 = p[5];
        if (nc > 6) row.s6 = p[6];
        if (nc > 7) row.s7 = p[7];
    }
    return row;
}

__device__ __forceinline__ uint64_t tab_count(const FormatTabArgs& a, const CountRow& row, uint64_t r, uint32_t term)
{
    const uint32_t pass = term >> 3, cls = term & 7;      // (the same for every lane)
    if (pass != 0) return a.counts[pass][r * a.nclass[pass] + cls];
    uint64_t v = row.s0;
    v = cls == 1 ? row.s1 : v; v = cls == 2 ? row.s2 : v; v = cls == 3 ? row.s3 : v; v = cls == 4 ? row.s4 : v;
    v = cls == 5 ? row.s5 : v; v = cls == 6 ? row.s6 : v; v = cls == 7 ? row.s7 : v;
    return v;
}

__device__ __forceinline__ uint64_t tab_term_sum(const FormatTabArgs& a, const TabColumn& c, const CountRow& row, uint64_t r)
{
    uint64_t sum = 0;
    for (uint32_t k = 0; k < c.nterm; k++) sum += tab_count(a, row, r, c.term[k]);
    return sum;
}

// The numeric columns of record r ("\t<number>" each), as fx2tab.go:127-143 prints them: their length,
// and with PUT their characters at buf.
template <bool PUT>
__device__ uint32_t tab_numbers(const FormatTabArgs& a, const CountRow& row, uint64_t r, uint64_t len, uint8_t* buf)
{
    uint32_t n = 0;
    uint64_t g = 0, cn = 0;
    bool have_gc = false;
    const bool small = len - 1 < 65535;                   // 1 <= len < 65536: fixed2_ratio_fast applies
    const uint32_t len32 = (uint32_t)len;
    const float rlen = small ? __builtin_amdgcn_rcpf((float)len32) : 0.0f;
    for (uint32_t ci = 0; ci < a.ncols; ci++) {
        const TabColumn& c = a.cols[ci];
        if (PUT) buf[n] = '\t';
        n++;
        if (c.kind == TAB_LENGTH) {
            const uint32_t nd = ndigits(len);
            if (PUT) put_decimal(len, nd, buf + n);
            n += nd;
            continue;
        }
        Fixed2 f;
        if (c.kind == TAB_CONTENT) {
            const uint64_t sum = tab_term_sum(a, c, row, r);
            if (!(small && sum <= len && fixed2_ratio_fast((uint32_t)sum, len32, rlen, f))) {
                const double v = len ? (double)sum / (double)len : 0.0;
                f = fixed2_of(v * 100);
            }
            n += fixed2_text<PUT>(f, buf + n);
            continue;
        }
        if (!have_gc) {
            g = tab_count(a, row, r, c.term[0]);
            cn = tab_count(a, row, r, c.term[1]);
            have_gc = true;
        }
        if (c.kind == TAB_GC && small && g + cn <= len && fixed2_ratio_fast((uint32_t)(g + cn), len32, rlen, f)) {
            n += fixed2_text<PUT>(f, buf + n);
            continue;
        }
        // two separate divisions, then the sum (fx2tab.go:128-133)
        const double gg = len ? (double)g / (double)len : 0.0, cc = len ? (double)cn / (double)len : 0.0;
        if (c.kind == TAB_GC) n += fixed2_text<PUT>(fixed2_of((gg + cc) * 100), buf + n);
        else n += fixed2_text<PUT>(fixed2_of((gg - cc) / (gg + cc) * 100), buf + n);
    }
    return n;
}

// The length of the numeric columns when every percentage among them can be had from one integer division
// (fixed2_ratio_fast: records of 1 to 65535 bases, no GC-skew column, no tie at the hundredths) -- `ok` says
// whether that held; if not, the value returned means nothing.  Straight-line code but for the column kinds,
// which are the same for every lane: a percentage q / 100 with q <= 10000 has 4 characters, 5 from 10.00, 6 at 100.00.
__device__ __forceinline__ uint32_t tab_numbers_size_fast(const FormatTabArgs& a, const CountRow& row, uint64_t r, uint64_t len64,
                                                          bool& ok)
{
    const uint32_t len = (uint32_t)len64;
    const float rlen = __builtin_amdgcn_rcpf((float)len);
    bool good = len64 - 1 < 65535;
    uint32_t n = 0;
    uint64_t gc = 0;
    bool have_gc = false;
    for (uint32_t ci = 0; ci < a.ncols; ci++) {
        const TabColumn& c = a.cols[ci];
        n++;
        if (c.kind == TAB_LENGTH) {
            n += 1u + (len >= 10u) + (len >= 100u) + (len >= 1000u) + (len >= 10000u);
            continue;
        }
        if (c.kind == TAB_GC_SKEW) { good = false; continue; }
        uint64_t sum;
        if (c.kind == TAB_CONTENT) sum = tab_term_sum(a, c, row, r);
        else {
            if (!have_gc) { gc = tab_count(a, row, r, c.term[0]) + tab_count(a, row, r, c.term[1]); have_gc = true; }
            sum = gc;
        }
        good = good && sum <= len64;
        const uint32_t N = (uint32_t)sum * 10000u;
        uint32_t q = (uint32_t)((float)N * rlen);
        uint32_t rem = N - q * len;
        const bool under = (int32_t)rem < 0, over = !under && rem >= len;
        q = under ? q - 1 : (over ? q + 1 : q);
        rem = under ? rem + len : (over ? rem - len : rem);
        good = good && 2u * rem != len;
        q += 2u * rem > len ? 1u : 0u;
        n += 4u + (q >= 1000u) + (q >= 10000u);
    }
    ok = good;
    return n;
}

// what follows the quality field of row r: the numeric columns, the alphabet column (fx2tab.go:145-147:
// a tab and the byte values present, ascending) and the line end
__device__ __forceinline__ void tab_row_tail(const FormatTabArgs& a, const CountRow& row, uint64_t r, uint64_t len, uint8_t* o)
{
    o += tab_numbers<true>(a, row, r, len, o);
    if (a.letters) {
        *o++ = '\t';
        for (uint32_t i = 0; i < 8; i++) {
            uint32_t w = a.letters[r * 8 + i];
            while (w) {
                const uint32_t bit = (uint32_t)__builtin_ctz(w);
                *o++ = (uint8_t)(32 * i + bit);
                w &= w - 1;
            }
        }
    }
    *o = '\n';
}

__global__ __launch_bounds__(FB) void k_tab_sizes(const FormatTabArgs a, uint64_t* __restrict__ size,
                                                  uint32_t* __restrict__ shown_name)
{
    for (uint64_t r = (uint64_t)blockIdx.x * FB + threadIdx.x; r < a.nrec; r += (uint64_t)gridDim.x * FB) {
        const CountRow row = load_count_row(a, r);
        const uint64_t len = a.offsets[r + 1] - a.offsets[r];
        uint32_t nl = a.name_len[r];
        if (a.only_id) nl = id_length(a.text + a.name_pos[r], nl);
        bool fast;
        uint32_t nn = tab_numbers_size_fast(a, row, r, len, fast);
        if (!fast) nn = tab_numbers<false>(a, row, r, len, nullptr);
        uint32_t na = 0;                                  // alphabet column: a tab and the letters present
        if (a.letters) {
            na = 1;
#pragma unroll
            for (int i = 0; i < 8; i++) na += (uint32_t)__builtin_popcount(a.letters[r * 8 + i]);
        }
        size[r] = (uint64_t)nl + 1 + (a.print_seq ? len : 0) + 1 + (a.print_qual ? len : 0) + nn + na + 1;
        shown_name[r] = nl;
    }
}

// Rows of as many records as fit the LDS stage, put together there by one wavefront and written out as
// aligned whole-wavefront rows (see k_fmt_write).  The numeric columns are formatted by one lane per
// record, all lanes busy; names, bases and qualities are copied by 16 lanes per record.
__global__ __launch_bounds__(FB) void k_tab_rows(const FormatTabArgs a, const uint64_t* __restrict__ size,
                                                 const uint64_t* __restrict__ pos, const uint32_t* __restrict__ shown_name,
                                                 const uint64_t* __restrict__ total, uint32_t* __restrict__ n_direct,
                                                 uint8_t* __restrict__ out, uint64_t out_capacity)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[FB / 64][FW_STAGE + 32];
    const uint64_t tot = *total;
    if (tot > out_capacity) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t R = fmt_records_per_step(tot, a.nrec);
    const uint64_t nsteps = (a.nrec + R - 1) / R;
    if (all_steps_direct(tot, a.nrec)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *n_direct = 1u;
        return;
    }
    const bool print_seq = a.print_seq, print_qual = a.print_qual;
    uint8_t* stage = s_stage[wave];
    for (uint64_t step = (uint64_t)blockIdx.x * (FB / 64) + wave; step < nsteps; step += (uint64_t)gridDim.x * (FB / 64)) {
        const uint64_t r0 = step * R;
        const uint32_t n = a.nrec - r0 < R ? (uint32_t)(a.nrec - r0) : R;
        // lane i holds the tables of record r0 + i
        uint64_t t_pos = 0, t_end = 0, t_s = 0, t_name = 0, t_len64 = 0;
        uint32_t t_nl = 0, t_len = 0;                     // t_len: for the copies (a printed field fits the stage)
        CountRow t_row = 0;
        if (lane < n) {
            const uint64_t r = r0 + lane;
            t_row = load_count_row(a, r);
            const uint64_t s = a.offsets[r], e = a.offsets[r + 1];
            t_pos = pos[r]; t_end = t_pos + size[r]; t_s = s; t_len64 = e - s; t_len = t_len64 > 0xffffffffull ? 0xffffffffu : (uint32_t)t_len64;
            t_nl = shown_name[r]; t_name = a.name_pos[r];
        }
        const uint64_t p0 = from_lane64(t_pos, 0), p1 = from_lane64(t_end, n - 1);
        const uint64_t a0 = p0 & ~15ull;
        if (p1 - a0 > FW_STAGE) {                          // wave-uniform
            if (lane == 0) *n_direct = 1u;                 // (every writer stores the same value)
            continue;
        }
        // (a row that fits the stage holds its bases and qualities: the record is shorter than the stage)
        if (lane < n) {
            const uint32_t at = (uint32_t)(t_pos - a0) + t_nl + 1 + (print_seq ? t_len : 0) + 1 + (print_qual ? t_len : 0);
            tab_row_tail(a, t_row, r0 + lane, t_len64, stage + at);
        }
        // the tabs by each record's own lane; the 16-byte pieces of names, bases and qualities of all rows dealt
        // over the 64 lanes, field by field (see write_step_flat)
        const uint32_t op = (uint32_t)(t_pos - a0);
        const uint32_t f_len = lane < n ? t_len : 0u, f_nl = lane < n ? t_nl : 0u;
        if (lane < n) {
            stage[op + t_nl] = '\t';
            stage[op + t_nl + 1 + (print_seq ? t_len : 0u)] = '\t';
        }
        const uint64_t s0 = from_lane64(t_s, 0);
        const uint32_t s_rel = (uint32_t)(t_s - s0);
        const Div32 no_wrap(0);
#pragma unroll 1
        for (int f = 0; f < 3; f++) {
            if ((f == 1 && !print_seq) || (f == 2 && !print_qual)) continue;
            const uint32_t PF = wave_max32(((f == 0 ? f_nl : f_len) + 15) >> 4);
            if (PF == 0) continue;
            flat_copy_field<FMT_ROUNDS>(stage, lane, n, PF, false, no_wrap, [&](uint32_t r, uint32_t& L, uint32_t& at, const uint8_t*& src) {
                const uint32_t op_r = (uint32_t)__shfl((int)op, (int)r), len_r = (uint32_t)__shfl((int)f_len, (int)r);
                const uint32_t nl_r = (uint32_t)__shfl((int)f_nl, (int)r), srel_r = (uint32_t)__shfl((int)s_rel, (int)r);
                const uint64_t name_r = f == 0 ? from_lane64(t_name, r) : 0;
                L = f == 0 ? nl_r : len_r;
                at = op_r + (f == 0 ? 0u : nl_r + 1 + (f == 2 ? (print_seq ? len_r : 0u) + 1 : 0u));
                src = f == 0 ? a.text + name_r : (f == 1 ? a.seq : a.qual) + s0 + srel_r;
            });
        }
        wave_lds_handover();
        const uint32_t nbytes = (uint32_t)(p1 - a0), head = (uint32_t)(p0 - a0);
        const uint32_t c_lo = head ? 16u : 0u, c_hi = nbytes & ~15u;
        for (uint32_t c = lane * 16; c < c_hi; c += 1024)
            if (c >= c_lo) *reinterpret_cast<uint4*>(out + a0 + c) = *reinterpret_cast<const uint4*>(stage + c);
        if (head && lane >= head && lane < 16 && lane < nbytes) out[a0 + lane] = stage[lane];
        if (c_hi < nbytes && (c_hi != 0 || !head) && lane < nbytes - c_hi) out[a0 + c_hi + lane] = stage[c_hi + lane];
        wave_lds_handover();
    }
}

// The rows of the steps k_tab_rows left out, straight to global memory: names, bases and qualities ...
__global__ __launch_bounds__(FB) void k_tab_write(const FormatTabArgs a, const uint64_t* __restrict__ size,
                                                  const uint64_t* __restrict__ pos,
                                                  const uint32_t* __restrict__ shown_name,
                                                  const uint64_t* __restrict__ total, const uint32_t* __restrict__ n_direct,
                                                  uint8_t* __restrict__ out, uint64_t out_capacity)
{
    constexpr uint32_t G = 16;
    const uint64_t tot = *total;
    if (*n_direct == 0 || tot > out_capacity) return;
    const uint32_t j = threadIdx.x & (G - 1);
    const uint32_t leader = (threadIdx.x & 63) & ~(G - 1);
    const uint32_t R = fmt_records_per_step(tot, a.nrec);
    const bool all_direct = all_steps_direct(tot, a.nrec);
    const Div32 no_wrap(0);
    for (uint64_t r = ((uint64_t)blockIdx.x * FB + threadIdx.x) / G; r < a.nrec; r += (uint64_t)gridDim.x * (FB / G)) {
        uint64_t m_pos = 0, m_s = 0, m_e = 0, m_name = 0;
        uint32_t m_nl = 0, m_take = 0;
        if (j == 0 && (all_direct || step_is_direct(r, R, a.nrec, pos, size))) {
            m_take = 1; m_pos = pos[r]; m_s = a.offsets[r]; m_e = a.offsets[r + 1]; m_nl = shown_name[r]; m_name = a.name_pos[r];
        }
        if (__shfl(m_take, leader) == 0) continue;
        uint8_t* o = out + from_lane64(m_pos, leader);
        const uint64_t s = from_lane64(m_s, leader), len = from_lane64(m_e, leader) - s;
        const uint32_t nl = __shfl(m_nl, leader);
        copy_field_direct<G>(o, a.text + from_lane64(m_name, leader), nl, false, no_wrap, j);
        if (j == 0) o[nl] = '\t';
        o += nl + 1;
        const bool is_long = len > SKH_LONG_T;           // bases and qualities: k_tab_write_long
        if (a.print_seq) { if (!is_long) copy_field_direct<G>(o, a.seq + s, (uint32_t)len, false, no_wrap, j); o += len; }
        if (j == 0) *o = '\t';
        o++;
        if (a.print_qual) { if (!is_long) copy_field_direct<G>(o, a.qual + s, (uint32_t)len, false, no_wrap, j); o += len; }
    }
}

// ... and the numeric columns and the line end, one thread per record
__global__ __launch_bounds__(FB) void k_tab_numbers(const FormatTabArgs a, const uint64_t* __restrict__ size,
                                                    const uint64_t* __restrict__ pos,
                                                    const uint32_t* __restrict__ shown_name,
                                                    const uint64_t* __restrict__ total, const uint32_t* __restrict__ n_direct,
                                                    uint8_t* __restrict__ out, uint64_t out_capacity)
{
    const uint64_t tot = *total;
    if (*n_direct == 0 || tot > out_capacity) return;
    const uint32_t R = fmt_records_per_step(tot, a.nrec);
    const bool all_direct = all_steps_direct(tot, a.nrec);
    for (uint64_t r = (uint64_t)blockIdx.x * FB + threadIdx.x; r < a.nrec; r += (uint64_t)gridDim.x * FB) {
        if (!all_direct && !step_is_direct(r, R, a.nrec, pos, size)) continue;
        const uint64_t len = a.offsets[r + 1] - a.offsets[r];
        tab_row_tail(a, load_count_row(a, r), r, len, out + pos[r] + shown_name[r] + 1 + (a.print_seq ? len : 0) + 1 + (a.print_qual ? len : 0));
    }
}

// Sequence and quality fields of the rows of records longer than SKH_LONG_T (windows of the
// sequence plane, as in k_fmt_write_long).
__global__ __launch_bounds__(FB) void k_tab_write_long(const FormatTabArgs a, const uint64_t* __restrict__ pos,
                                                       const uint32_t* __restrict__ shown_name,
                                                       const uint64_t* __restrict__ total, uint8_t* __restrict__ out,
                                                       uint64_t out_capacity)
{
    if (*total > out_capacity) return;
    const uint64_t plane = a.offsets[a.nrec];
    const uint64_t ntiles = (plane + SKH_TILE - 1) / SKH_TILE;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t t0 = tile * SKH_TILE;
        const uint64_t t1 = t0 + SKH_TILE < plane ? t0 + SKH_TILE : plane;
        const uint64_t r_lo = find_record(a.offsets, a.nrec, t0);
        const uint64_t r_hi = a.offsets[r_lo + 1] > t1 - 1 ? r_lo : find_record(a.offsets, a.nrec, t1 - 1);
#pragma unroll 1
        for (int c = 0; c < 2; c++) {
            if (c == 1 && r_hi == r_lo) break;
            const uint64_t r = c == 0 ? r_lo : r_hi;
            const uint64_t s = a.offsets[r], e = a.offsets[r + 1], len = e - s;
            if (len <= SKH_LONG_T) continue;                          // block-uniform
            const uint64_t k0 = (s > t0 ? s : t0) - s, k1 = (e < t1 ? e : t1) - s;
            uint8_t* o = out + pos[r] + shown_name[r] + 1;
            if (a.print_seq) { copy_wrapped_range(o, a.seq + s, len, 0, k0, k1); o += len; }
            o++;
            if (a.print_qual) copy_wrapped_range(o, a.qual + s, len, 0, k0, k1);
        }
    }
}

__global__ void k_fmt_finish(const uint64_t* __restrict__ total, uint64_t out_capacity, FormatResult* __restrict__ res)
{
    res->out_bytes = *total;
    res->status = *total > out_capacity ? FORMAT_ECAPACITY : FORMAT_OK;
    res->pad = 0;
}

static inline uint64_t up256(uint64_t x) { return (x + 255) & ~255ull; }

}  // namespace

uint64_t scan64_partials(uint64_t n) { return (n + FSCAN_TILE - 1) / FSCAN_TILE + 1; }

hipError_t launch_scan64(const uint64_t* v, uint64_t n, uint64_t* out, uint64_t* part, uint64_t* total, hipStream_t s)
{
    const uint64_t npart = (n + FSCAN_TILE - 1) / FSCAN_TILE ? (n + FSCAN_TILE - 1) / FSCAN_TILE : 1;
    hipLaunchKernelGGL(k_scan_partials64, dim3((uint32_t)npart), dim3(FB), 0, s, v, n, part);
    hipLaunchKernelGGL(k_scan_top64, dim3(1), dim3(1024), 0, s, part, npart, total);
    hipLaunchKernelGGL(k_scan_apply64, dim3((uint32_t)npart), dim3(FB), 0, s, v, n, part, out);
    return hipGetLastError();
}

uint64_t format_workspace_size(uint64_t nrec)
{
    const uint64_t npart = (nrec + FSCAN_TILE - 1) / FSCAN_TILE + 1;
    return up256(8) + up256(nrec * 8) + up256(nrec * 8) + up256(nrec * 4) + up256(npart * 8);
}

hipError_t launch_format_seq(const FormatSeqArgs& a, uint8_t* out, uint64_t out_capacity, void* workspace,
                             FormatResult* res, hipStream_t s)
{
    uint8_t* ws = static_cast<uint8_t*>(workspace);
    uint64_t* total = reinterpret_cast<uint64_t*>(ws); ws += up256(8);
    uint64_t* size = reinterpret_cast<uint64_t*>(ws); ws += up256(a.nrec * 8);
    uint64_t* pos = reinterpret_cast<uint64_t*>(ws); ws += up256(a.nrec * 8);
    uint32_t* shown = reinterpret_cast<uint32_t*>(ws); ws += up256(a.nrec * 4);
    uint64_t* part = reinterpret_cast<uint64_t*>(ws);
    const uint64_t npart = (a.nrec + FSCAN_TILE - 1) / FSCAN_TILE ? (a.nrec + FSCAN_TILE - 1) / FSCAN_TILE : 1;
    auto capped = [](uint64_t b) { return (uint32_t)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); };
    uint32_t* n_long = reinterpret_cast<uint32_t*>(total + 1);       // second word of the 256-byte head of the workspace
    hipError_t e = hipMemsetAsync(total, 0, 16, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fmt_sizes, dim3(capped((a.nrec + FB - 1) / FB)), dim3(FB), 0, s, a, size, shown, n_long);
    hipLaunchKernelGGL(k_scan_partials64, dim3((uint32_t)npart), dim3(FB), 0, s, size, a.nrec, part);
    hipLaunchKernelGGL(k_scan_top64, dim3(1), dim3(1024), 0, s, part, npart, total);
    hipLaunchKernelGGL(k_scan_apply64, dim3((uint32_t)npart), dim3(FB), 0, s, size, a.nrec, part, pos);
    uint32_t* n_direct = n_long + 1;                                 // third word
    hipLaunchKernelGGL(k_fmt_write, dim3(capped((a.nrec + 7) / 8)), dim3(FB), 0, s, a, size, pos, shown, total, n_direct,
                       out, out_capacity);
    hipLaunchKernelGGL(k_fmt_write_direct, dim3(capped((a.nrec * 16 + FB - 1) / FB)), dim3(FB), 0, s, a, size, pos, shown, total,
                       n_direct, out, out_capacity);
    if (a.nrec != 0 && (a.print_seq || a.print_qual))
        hipLaunchKernelGGL(k_fmt_write_long, dim3(2048), dim3(FB), 0, s, a, size, pos, shown, total, n_long, out, out_capacity);
    hipLaunchKernelGGL(k_fmt_finish, dim3(1), dim3(1), 0, s, total, out_capacity, res);
    return hipGetLastError();
}

hipError_t launch_format_tab(const FormatTabArgs& a, uint8_t* out, uint64_t out_capacity, void* workspace,
                             FormatResult* res, hipStream_t s)
{
    uint8_t* ws = static_cast<uint8_t*>(workspace);
    uint64_t* total = reinterpret_cast<uint64_t*>(ws); ws += up256(8);
    uint64_t* size = reinterpret_cast<uint64_t*>(ws); ws += up256(a.nrec * 8);
    uint64_t* pos = reinterpret_cast<uint64_t*>(ws); ws += up256(a.nrec * 8);
    uint32_t* shown = reinterpret_cast<uint32_t*>(ws); ws += up256(a.nrec * 4);
    uint64_t* part = reinterpret_cast<uint64_t*>(ws);
    const uint64_t npart = (a.nrec + FSCAN_TILE - 1) / FSCAN_TILE ? (a.nrec + FSCAN_TILE - 1) / FSCAN_TILE : 1;
    auto capped = [](uint64_t b) { return (uint32_t)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); };
    hipError_t e = hipMemsetAsync(total, 0, 16, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_tab_sizes, dim3(capped((a.nrec + FB - 1) / FB)), dim3(FB), 0, s, a, size, shown);
    hipLaunchKernelGGL(k_scan_partials64, dim3((uint32_t)npart), dim3(FB), 0, s, size, a.nrec, part);
    hipLaunchKernelGGL(k_scan_top64, dim3(1), dim3(1024), 0, s, part, npart, total);
    hipLaunchKernelGGL(k_scan_apply64, dim3((uint32_t)npart), dim3(FB), 0, s, size, a.nrec, part, pos);
    uint32_t* n_direct = reinterpret_cast<uint32_t*>(total + 1) + 1;     // third word of the workspace head (zeroed above)
    hipLaunchKernelGGL(k_tab_rows, dim3(capped((a.nrec + 7) / 8)), dim3(FB), 0, s, a, size, pos, shown, total, n_direct, out,
                       out_capacity);
    hipLaunchKernelGGL(k_tab_write, dim3(capped((a.nrec * 16 + FB - 1) / FB)), dim3(FB), 0, s, a, size, pos, shown, total,
                       n_direct, out, out_capacity);
    if (a.nrec != 0 && (a.print_seq || a.print_qual))
        hipLaunchKernelGGL(k_tab_write_long, dim3(2048), dim3(FB), 0, s, a, pos, shown, total, out, out_capacity);
    hipLaunchKernelGGL(k_tab_numbers, dim3(capped((a.nrec + FB - 1) / FB)), dim3(FB), 0, s, a, size, pos, shown, total, n_direct,
                       out, out_capacity);
    hipLaunchKernelGGL(k_fmt_finish, dim3(1), dim3(1), 0, s, total, out_capacity, res);
    return hipGetLastError();
}

}  // namespace skh
