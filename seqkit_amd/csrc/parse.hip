// parse.hip -- FASTA / FASTQ text -> CSR batch on the GPU (gfx950).
//
// The step BEFORE the per-record hot path: the reference's commands obtain records from
// fastx.Reader (call sites seqkit/cmd/seq.go:115-127, fx2tab.go:96-106, stat.go:87-104), a
// serial byte scanner.  Here a block of file text that already lies in HBM is turned into the
// batch layout of the other kernels -- sequence plane, quality plane, offsets, name spans --
// by data-parallel passes, all of them streaming and HBM-bound:
//
//   1. newline index   per 16-KiB tile, in one pass over the text: count of '\n' and their entries, tile by tile;
//                      scan of the tile counts
//   2. line table      per line: start, length without trailing '\r', head flag, and the
//                      bytes it contributes to the sequence plane
//   3. scan            exclusive sums over lines of (contribution, head flag): a line's place
//                      in the plane and the record it belongs to
//   4. emit            offsets[] and name spans at head lines; copy of every sequence (and
//                      quality) line to its place
//
// Rules (the same the host reader of this repository applies, fastx_reader.cpp):
//   FASTA  a record starts at a line whose first byte is '>', and runs to the next such line;
//          line ends ("\n", "\r\n") and blank lines are dropped from the sequence, nothing else
//          is; the name is the head line without '>'.
//   FASTQ  four lines per record: '@'name, sequence, '+'..., quality of the same length.
//          Blank lines are accepted only after the last record.  (Multi-line FASTQ cannot be
//          told apart from data without a serial scan -- '@' may start a quality line -- and is
//          reported as a format error by this parser.)
// A block that is not the last of its file ends in an incomplete record: `consumed` tells the
// caller where that record starts, so that the next block can begin there.
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "device_util.h"

namespace skh {

namespace {

constexpr int PB = 256;                      // threads per workgroup
constexpr uint64_t LONG_LINE = 32768;         // lines longer than this are copied window-wise
constexpr int NLU = 4;                       // 16-byte chunks per thread in the newline passes
constexpr uint32_t TILE = PB * 16 * NLU;     // text bytes per workgroup in the newline passes

// An entry of the newline index: the position of a '\n', with two facts about its
// neighbourhood packed into the top bits while the bytes are at hand -- whether the byte
// before it is '\r', and the first byte of the following line -- so that the line table is
// built without touching the text again.
constexpr uint64_t NL_POS = (1ull << 55) - 1;
constexpr int NL_NEXT_SHIFT = 55;            // 8 bits: first byte after the newline (0 at the end of the text)
constexpr uint64_t NL_CR = 1ull << 63;       // the line ends in "\r\n"
constexpr int SCAN_ITEMS = 8;                // items per thread in the line scans
constexpr uint32_t SCAN_TILE = PB * SCAN_ITEMS;

struct LineSum { uint64_t bytes; uint32_t heads; };

__device__ __forceinline__ uint32_t nl_mask16(uint4 v)       // bit k set: byte k is '\n'
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t m = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t x = w[d] ^ 0x0a0a0a0au;              // zero byte where '\n'
        // exact zero-byte detector: 0x80 in every byte of x that is zero, nothing elsewhere
        const uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);
        // the four flags (bits 7, 15, 23, 31) to bits 0..3: two folds instead of four extractions
        const uint32_t t = z >> 7, u = t | (t >> 7);
        m |= ((u | (u >> 14)) & 0xfu) << (4 * d);
    }
    return m;
}

// block-wide exclusive scan of one uint32 per thread; returns the exclusive prefix, *total = sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* total, uint32_t* s_wave)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d);
        if ((int)lane >= d) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < PB / 64; w++) {
        if (w < (int)wave) base += s_wave[w];
        sum += s_wave[w];
    }
    __syncthreads();
    *total = sum;
    return base + inc - v;
}

// ---- pass 1: newline index ---------------------------------------------------------------

__device__ __forceinline__ uint32_t nl_mask_at(const uint8_t* __restrict__ text, uint64_t n, uint64_t o)
{
    if (o >= n) return 0;
    uint32_t m = nl_mask16(*reinterpret_cast<const uint4*>(text + o));
    if (o + 16 > n) m &= (1u << (n - o)) - 1u;
    return m;
}

// number of '\n' among the 16 bytes: the zero-byte detector of nl_mask16 without putting the mask together
__device__ __forceinline__ uint32_t nl_count16(uint4 v)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t other = 0;                                     // bytes that are not '\n'
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t x = w[d] ^ 0x0a0a0a0au;
        other += __builtin_popcount((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u);
    }
    return 16u - other;
}

// A tile's newlines as the line table wants them, in ONE pass over the text: the tile's count, and the
// entries themselves in the tile's own stretch of a table with room for TILE entries per tile (a tile
// cannot hold more newlines than bytes), 32 bits each: position within the tile, the CR flag and the
// first byte of the next line.  Where an entry belongs among ALL newlines is only known after the scan of
// the tile counts; the line table reads its entries tile by tile and does not need to know.  (Counting
// first and scattering behind the scanned bases read the text twice.  The table is four times the size of
// the text in address space; only the entries that exist are ever touched.)
constexpr int NLT_POS_BITS = 14;                              // TILE = 2^14
static_assert(TILE == (1u << NLT_POS_BITS), "tile-relative positions");
constexpr uint32_t NLT_CR = 1u << 22;

__device__ __forceinline__ uint64_t nl_entry64(uint32_t e32, uint64_t t0)
{
    return (t0 + (e32 & (TILE - 1))) | ((uint64_t)((e32 >> NLT_POS_BITS) & 0xffu) << NL_NEXT_SHIFT) | ((e32 & NLT_CR) ? NL_CR : 0);
}

__global__ __launch_bounds__(PB) void k_nl_index(const uint8_t* __restrict__ text, uint64_t n,
                                                 uint32_t* __restrict__ tile_count, uint64_t* __restrict__ tile_last,
                                                 uint32_t* __restrict__ nl_tile)
{
    static_assert(NLU == 4 && PB * 16 < 65536, "four 16-bit fields");
    __shared__ uint64_t s_wave[PB / 64];
    __shared__ uint4 s_text[TILE / 16];                   // the tile: the bytes next to a newline are looked up here
    const uint8_t* s_byte = reinterpret_cast<const uint8_t*>(s_text);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t t0 = (uint64_t)blockIdx.x * TILE;
    uint32_t m[NLU];
    {
        uint4 v[NLU];
#pragma unroll
        for (int u = 0; u < NLU; u++) {
            const uint64_t o = t0 + (uint64_t)(u * PB + threadIdx.x) * 16;
            v[u] = o < n ? *reinterpret_cast<const uint4*>(text + o) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < NLU; u++) {
            const uint64_t o = t0 + (uint64_t)(u * PB + threadIdx.x) * 16;
            s_text[u * PB + threadIdx.x] = v[u];
            m[u] = nl_mask16(v[u]);
            if (o + 16 > n) m[u] &= o < n ? (1u << (n - o)) - 1u : 0u;
        }
    }
    uint64_t mine = 0;
#pragma unroll
    for (int u = 0; u < NLU; u++) mine |= (uint64_t)__builtin_popcount(m[u]) << (16 * u);
    uint64_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t t = ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(inc >> 32), d) << 32) | (uint32_t)__shfl_up((int)(uint32_t)inc, d);
        if ((int)lane >= d) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint64_t before = 0, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < PB / 64; w++) {
        const uint64_t x = s_wave[w];
        if (w < wave) before += x;
        all += x;
    }
    const uint64_t ex = before + inc - mine;             // per row: newlines of the row before this thread's chunk
    uint32_t* mine_out = nl_tile + (uint64_t)blockIdx.x * TILE;
    const uint32_t count = (uint32_t)((all & 0xffffu) + ((all >> 16) & 0xffffu) + ((all >> 32) & 0xffffu) + ((all >> 48) & 0xffffu));
    // where the thread's first entry of each of its four chunks goes: the rows before, and the row's threads before
    uint32_t kb[NLU], run = 0;
#pragma unroll
    for (int u = 0; u < NLU; u++) {
        kb[u] = run + (uint32_t)((ex >> (16 * u)) & 0xffffu);
        run += (uint32_t)((all >> (16 * u)) & 0xffffu);
    }
    // ONE loop over the newlines of all four chunks (a lane finds a newline in one chunk out of five or so: four
    // loops run as long as the busiest lane of each chunk row, one loop as long as the busiest lane overall)
    const uint64_t mall = (uint64_t)m[0] | ((uint64_t)m[1] << 16) | ((uint64_t)m[2] << 32) | ((uint64_t)m[3] << 48);
    uint64_t mm = mall;
    while (mm) {
        const uint32_t bit = (uint32_t)__builtin_ctzll(mm);
        mm &= mm - 1;
        const uint32_t u = bit >> 4, b = bit & 15u;
        const uint32_t lower = (uint32_t)(mall >> (bit & 48u)) & ((1u << b) - 1u);       // the chunk's newlines before this one
        const uint32_t k = (u == 0 ? kb[0] : (u == 1 ? kb[1] : (u == 2 ? kb[2] : kb[3]))) + (uint32_t)__builtin_popcount(lower);
        const uint32_t q = ((u * PB + threadIdx.x) << 4) + b;              // position in the tile
        const uint64_t p = t0 + q;                          // the neighbours outside the tile come from memory
        const uint32_t prev = q > 0 ? s_byte[q - 1] : (p > 0 ? text[p - 1] : 0);
        const uint32_t next = p + 1 >= n ? 0 : (q + 1 < TILE ? s_byte[q + 1] : text[p + 1]);
        const uint32_t e32 = q | (next << NLT_POS_BITS) | (prev == '\r' ? NLT_CR : 0u);
        mine_out[k] = e32;
        if (k + 1 == count) tile_last[blockIdx.x] = nl_entry64(e32, t0);
    }
    if (threadIdx.x == 0) tile_count[blockIdx.x] = count;
}

// Exclusive scan of the ntile uint32 newline counts into uint64 bases, in three small launches: sums of 4096
// counts per workgroup, a scan of those sums by one workgroup (which also sets up the state of the call: the
// number of newlines is known there), and the bases from the scanned sums.  (ONE workgroup over all counts has
// the bandwidth and the latency of one compute unit: rounds of 16,384 counts took 0.044 ms on 83 K tiles, rounds
// of 4,096 0.060 ms, one round with 81 consecutive counts per thread 0.055 ms.)
constexpr uint32_t TS_PER = 16, TS_TILE = PB * TS_PER;      // counts per thread and per workgroup

__device__ __forceinline__ void tile_counts16(const uint32_t* __restrict__ cnt, uint64_t ntile, uint64_t i, uint32_t (&c)[TS_PER])
{
    if (i + TS_PER <= ntile) {                                       // (cnt is 256-byte aligned, i a multiple of 16)
#pragma unroll
        for (uint32_t k = 0; k < TS_PER; k += 4) {
            const uint4 v = *reinterpret_cast<const uint4*>(cnt + i + k);
            c[k] = v.x; c[k + 1] = v.y; c[k + 2] = v.z; c[k + 3] = v.w;
        }
    } else {
#pragma unroll
        for (uint32_t k = 0; k < TS_PER; k++) c[k] = i + k < ntile ? cnt[i + k] : 0u;
    }
}

__global__ __launch_bounds__(PB) void k_tiles_partials(const uint32_t* __restrict__ cnt, uint64_t ntile, uint64_t* __restrict__ tpart)
{
    __shared__ uint64_t s_wave[PB / 64];
    uint32_t c[TS_PER];
    tile_counts16(cnt, ntile, (uint64_t)blockIdx.x * TS_TILE + (uint64_t)threadIdx.x * TS_PER, c);
    uint64_t mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < TS_PER; k++) mine += c[k];
    for (int d = 1; d < 64; d <<= 1) mine += __shfl_xor((unsigned long long)mine, d);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (uint32_t w = 0; w < PB / 64; w++) t += s_wave[w];
        tpart[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(PB) void k_tiles_apply(const uint32_t* __restrict__ cnt, uint64_t ntile, const uint64_t* __restrict__ tpart,
                                                    uint64_t* __restrict__ base)
{
    __shared__ uint64_t s_wave[PB / 64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t i = (uint64_t)blockIdx.x * TS_TILE + (uint64_t)threadIdx.x * TS_PER;
    uint32_t c[TS_PER];
    tile_counts16(cnt, ntile, i, c);
    uint64_t mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < TS_PER; k++) mine += c[k];
    uint64_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t t = __shfl_up((unsigned long long)inc, d);
        if ((int)lane >= d) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint64_t run = tpart[blockIdx.x] + inc - mine;
    for (uint32_t w = 0; w < wave; w++) run += s_wave[w];
    if (i + TS_PER <= ntile) {
#pragma unroll
        for (uint32_t k = 0; k < TS_PER; k += 2) {
            const uint64_t r0 = run, r1 = run + c[k];
            *reinterpret_cast<uint4*>(base + i + k) = make_uint4((uint32_t)r0, (uint32_t)(r0 >> 32), (uint32_t)r1, (uint32_t)(r1 >> 32));
            run = r1 + c[k + 1];
        }
    } else {
#pragma unroll
        for (uint32_t k = 0; k < TS_PER; k++) {
            if (i + k < ntile) base[i + k] = run;
            run += c[k];
        }
    }
}

// ---- pass 2: line table ------------------------------------------------------------------

struct ParseState {            // device scratch shared by the passes of one call
    uint64_t n_nl;             // newlines in the text
    uint64_t nlines;           // lines considered: newline-terminated ones, + an unterminated last line when final
    unsigned long long err_line;           // first line that breaks the format (min), ~0 = none
    unsigned long long last_nonblank;      // last non-blank line (max); valid if any_nonblank
    unsigned long long first_blank_head;   // FASTQ: first blank line in a head position (min), ~0 = none
    uint32_t any_nonblank;
    uint32_t status;
    uint32_t n_long;           // lines longer than LONG_LINE (copied window-wise)
    uint32_t pad;
    LineSum total;             // totals of the line scan
    uint64_t nrec, seq_bytes, consumed;    // the decision
    uint64_t nrec_cap;                     // complete records at most (known before the line scan), capped at the tables' size
    uint64_t first_head, last_head;        // FASTA: the first and the last head line (~0 = none), noted by the line scan
    uint64_t copy_lo, copy_hi;             // FASTA: the sequence lines i of the block's records are those with copy_lo < i < copy_hi
};

// One workgroup: the sums of k_tiles_partials scanned in place (exclusive), the number of newlines, and with it
// the state every later pass starts from.
__global__ __launch_bounds__(1024) void k_tiles_top(uint64_t* __restrict__ tpart, uint64_t ntpart, uint64_t ntile,
                                                    uint64_t* __restrict__ tile_base, ParseState* st, uint64_t n, int final,
                                                    const uint8_t* __restrict__ text, uint64_t max_lines)
{
    __shared__ uint64_t s_wave[16];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t carry = 0;
    for (uint64_t i0 = 0; i0 < ntpart; i0 += 1024) {
        const uint64_t i = i0 + threadIdx.x;
        const uint64_t mine = i < ntpart ? tpart[i] : 0;
        uint64_t inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t t = __shfl_up((unsigned long long)inc, d);
            if ((int)lane >= d) inc += t;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint64_t before = carry, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16; w++) {
            if (w < wave) before += s_wave[w];
            all += s_wave[w];
        }
        __syncthreads();
        if (i < ntpart) tpart[i] = before + inc - mine;
        carry += all;
    }
    if (threadIdx.x != 0) return;
    const uint64_t nl = carry;
    tile_base[ntile] = nl;
    st->n_nl = nl;
    uint64_t lines = nl;
    if (final && n != 0 && text[n - 1] != '\n') lines++;          // unterminated last line
    if (lines > max_lines) lines = max_lines;                      // reported as capacity error later
    st->nlines = lines;
    st->err_line = ~0ull;
    st->last_nonblank = 0;
    st->first_blank_head = ~0ull;
    st->any_nonblank = 0;
    st->n_long = 0;
    st->total.bytes = 0;
    st->total.heads = 0;
    st->first_head = ~0ull; st->last_head = ~0ull;
    st->copy_lo = ~0ull; st->copy_hi = 0;
}

// the three launches
static void launch_tile_scan(const uint32_t* tile_count, uint64_t ntile, uint64_t* tile_base, uint64_t* tpart, ParseState* st,
                             uint64_t n, int final, const uint8_t* text, uint64_t max_lines, hipStream_t s)
{
    const uint64_t ntpart = (ntile + TS_TILE - 1) / TS_TILE;
    if (ntpart) hipLaunchKernelGGL(k_tiles_partials, dim3((uint32_t)ntpart), dim3(PB), 0, s, tile_count, ntile, tpart);
    hipLaunchKernelGGL(k_tiles_top, dim3(1), dim3(1024), 0, s, tpart, ntpart, ntile, tile_base, st, n, final, text, max_lines);
    if (ntpart) hipLaunchKernelGGL(k_tiles_apply, dim3((uint32_t)ntpart), dim3(PB), 0, s, tile_count, ntile, tpart, tile_base);
}

// Line i: its start, its length without the trailing '\r's, and its first byte (0 if empty).
// `before` and `e` are the newline entries i - 1 and i (read by the caller; unused where there is none).
__device__ __forceinline__ void line_span(const uint8_t* __restrict__ text, uint64_t n, uint64_t before, uint64_t e,
                                          uint64_t n_nl, uint64_t i, uint64_t& ls, uint32_t& len, uint8_t& first)
{
    ls = i ? (before & NL_POS) + 1 : 0;
    uint64_t le;
    bool cr;
    if (i < n_nl) {
        le = e & NL_POS;
        cr = (e & NL_CR) != 0;
    } else {                                             // the unterminated last line
        le = n;
        cr = le > ls && text[le - 1] == '\r';
    }
    if (cr) { le--; while (le > ls && text[le - 1] == '\r') le--; }
    if (le < ls) le = ls;
    len = (uint32_t)(le - ls);            // lines are shorter than 4 GiB
    first = len ? (i ? (uint8_t)((before >> NL_NEXT_SHIFT) & 0xff) : text[0]) : 0;
}

// the entry of the last newline before tile t (0: none): of the nearest earlier tile that has one
__device__ __forceinline__ uint64_t newline_before_tile(uint64_t t, const uint32_t* __restrict__ tile_count,
                                                        const uint64_t* __restrict__ tile_last)
{
    while (t > 0) { t--; if (tile_count[t]) return tile_last[t]; }
    return 0;
}

constexpr uint32_t LT_TILES = 8;             // tiles of newline entries a workgroup of the line table takes at a time

// POS: the type of the tables of positions in the text and in the planes (line_start, line_dst) -- 32 bits for a
// text of less than 4 GiB, which every position then fits
template <bool FASTQ, typename POS>
__global__ __launch_bounds__(PB) void k_line_table(const uint8_t* __restrict__ text, uint64_t n, uint64_t ntile,
                                                   const uint32_t* __restrict__ nl_tile, const uint32_t* __restrict__ tile_count,
                                                   const uint64_t* __restrict__ tile_base, const uint64_t* __restrict__ tile_last,
                                                   ParseState* st, uint64_t* __restrict__ nlpos,
                                                   POS* __restrict__ line_start, uint32_t* __restrict__ line_len,
                                                   uint8_t* __restrict__ line_head, int raw_first,
                                                   uint64_t max_records, uint64_t* __restrict__ name_pos, uint32_t* __restrict__ name_len)
{
    __shared__ uint64_t s_base[LT_TILES + 1], s_before[LT_TILES];
    const uint64_t nlines = st->nlines, n_nl = st->n_nl;
    // per thread, then per wavefront, then one atomic each: the flags of the whole text
    unsigned long long err = ~0ull, blank_head = ~0ull, last = 0;
    bool any = false;
    // Line i ends at newline i.  The newlines are read from where k_nl_index left them, tile by tile (entry k of
    // tile t is newline tile_base[t] + k), LT_TILES tiles at a time: the lines of such a group are consecutive, a
    // thread finds the tile of its line among the group's bases.  They leave as the dense table of 64-bit entries
    // the later passes look lines up in.  The unterminated last line, if any, is line n_nl.
    const uint64_t ngroup = (ntile + LT_TILES - 1) / LT_TILES;
    // (from the last group down: the workgroups that run first then hold the highest line numbers, and the others
    // find in st->last_nonblank a value they cannot raise -- see the atomics at the end)
    for (uint64_t gr = blockIdx.x; gr <= ngroup; gr += gridDim.x) {
        const uint64_t g = ngroup - gr;
        const uint64_t T0 = g * LT_TILES;
        __syncthreads();
        if (threadIdx.x <= LT_TILES) { const uint64_t t = T0 + threadIdx.x; s_base[threadIdx.x] = tile_base[t < ntile ? t : ntile]; }
        // (the newline before a tile's first: looked up once per tile here, not by the thread of that line inside the loop)
        if (threadIdx.x >= 64 && threadIdx.x < 64 + LT_TILES) {
            const uint64_t t = T0 + (threadIdx.x - 64);
            s_before[threadIdx.x - 64] = newline_before_tile(t < ntile ? t : ntile, tile_count, tile_last);
        }
        __syncthreads();
        uint64_t i_lo = s_base[0], i_hi = s_base[LT_TILES];
        if (g == ngroup) { i_lo = n_nl; i_hi = nlines > n_nl ? n_nl + 1 : n_nl; }     // the line without a newline
        if (i_hi > nlines) i_hi = nlines;
        constexpr int LB = 4;                             // lines per thread whose entries are loaded before the first is used
        for (uint64_t ib = i_lo + threadIdx.x; ib < i_hi; ib += (uint64_t)PB * LB) {
            uint32_t ent_here[LB], ent_before[LB], w_of[LB], k_of[LB];
#pragma unroll
            for (int j = 0; j < LB; j++) {
                const uint64_t i = ib + (uint64_t)j * PB;
                ent_here[j] = 0; ent_before[j] = 0; w_of[j] = 0; k_of[j] = 0;
                if (i < i_hi && i < n_nl) {
                    uint32_t w = 0;
#pragma unroll
                    for (uint32_t x = 1; x < LT_TILES; x++) w += s_base[x] <= i ? 1u : 0u;  // tile T0 + w holds newline i
                    const uint32_t k = (uint32_t)(i - s_base[w]);
                    const uint32_t* ent = nl_tile + (T0 + w) * TILE;
                    ent_here[j] = ent[k];
                    if (k) ent_before[j] = ent[k - 1];
                    w_of[j] = w; k_of[j] = k;
                }
            }
#pragma unroll
            for (int j = 0; j < LB; j++) {
                const uint64_t i = ib + (uint64_t)j * PB;
                if (i >= i_hi) break;
                uint64_t e_here = 0, e_before;
                if (i < n_nl) {
                    const uint64_t t0 = (T0 + w_of[j]) * TILE;
                    e_here = nl_entry64(ent_here[j], t0);
                    e_before = k_of[j] ? nl_entry64(ent_before[j], t0) : s_before[w_of[j]];
                    if (i + 1 == nlines) nlpos[i] = e_here;      // (only the last line's newline is ever looked up again)
                } else {
                    e_before = s_before[0];                  // (T0 >= ntile here: every slot holds the text's last newline)
                }
                uint64_t ls; uint32_t len; uint8_t c;
                line_span(text, n, e_before, e_here, n_nl, i, ls, len, c);
                bool head, bad = false;
                if (!FASTQ) {
                    head = c == '>';
                } else {
                    const uint32_t role = (uint32_t)(i & 3);
                    head = role == 0;
                    if (role == 0) {
                        if (len == 0) blank_head = i < blank_head ? i : blank_head;
                        else bad = c != '@';
                    } else if (role == 2) {
                        bad = c != '+';
                    } else if (role == 1) {
                        bad = len != 0 && c == '+';          // the reader takes such a line for the '+' line: not four-line text
                    }                                        // role 3 (equal lengths) is checked by k_line_partials
                }
                if (len) { last = i > last ? i : last; any = true; }
                if (len > LONG_LINE) atomicAdd(&st->n_long, 1u);
                if (bad && i < err) err = i;
                line_start[i] = (POS)ls;
                line_len[i] = len;
                if (!FASTQ) line_head[i] = raw_first ? c : (head ? 1 : 0);     // (raw_first: the line's first byte, for the wrapped-FASTQ passes)
                // four-line FASTQ: line 4 r is the head of record r, whatever comes of the block (entries beyond the
                // records the block ends up with are never looked at)
                if (FASTQ && head && (i >> 2) < max_records) { name_pos[i >> 2] = ls + 1; name_len[i >> 2] = len - 1; }
            }
        }
    }
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long e2 = __shfl_xor(err, d), b2 = __shfl_xor(blank_head, d), l2 = __shfl_xor(last, d);
        err = e2 < err ? e2 : err;
        blank_head = b2 < blank_head ? b2 : blank_head;
        last = l2 > last ? l2 : last;
    }
    const bool wave_any = __any(any);
    // same-address atomics serialise in L2: one set per workgroup, from a small grid
    __shared__ unsigned long long s_err[PB / 64], s_blank[PB / 64], s_last[PB / 64];
    __shared__ uint32_t s_any[PB / 64];
    if ((threadIdx.x & 63) == 0) {
        const uint32_t w = threadIdx.x >> 6;
        s_err[w] = err; s_blank[w] = blank_head; s_last[w] = last; s_any[w] = wave_any ? 1u : 0u;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0;
        for (int w = 0; w < PB / 64; w++) {
            err = s_err[w] < err ? s_err[w] : err;
            blank_head = s_blank[w] < blank_head ? s_blank[w] : blank_head;
            last = s_last[w] > last ? s_last[w] : last;
            a |= s_any[w];
        }
        if (err != ~0ull) atomicMin(&st->err_line, err);
        if (blank_head != ~0ull) atomicMin(&st->first_blank_head, blank_head);
        // Same-address atomics serialise in L2 at some 15 ns each: 2,048 workgroups of them were half of this kernel.
        // The maximum only grows, so a workgroup whose value is not above what it reads leaves it alone (a stale read
        // costs an atomic, never a result; the reads go past the die's L2, which would keep its first answer).
        if (a) {
            if (last > __hip_atomic_load(&st->last_nonblank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ||
                !__hip_atomic_load(&st->any_nonblank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                atomicMax(&st->last_nonblank, last);
                st->any_nonblank = 1;
            }
        }
    }
}

// ---- pass 3: exclusive scan over lines of (bytes contributed, head flag) -------------------

template <bool FASTQ>
__device__ __forceinline__ LineSum line_item(const uint32_t* __restrict__ line_len,
                                             const uint8_t* __restrict__ line_head, uint64_t i, uint64_t nlines)
{
    LineSum v{0, 0};
    if (i < nlines) {
        const bool head = FASTQ ? (i & 3) == 0 : line_head[i] != 0;      // (four-line FASTQ: every fourth line, no table)
        v.heads = head ? 1u : 0u;
        if (FASTQ) v.bytes = (i & 3) == 1 ? line_len[i] : 0;
        else v.bytes = head ? 0 : line_len[i];
    }
    return v;
}

// The SCAN_ITEMS consecutive lines of a thread, from `base` (a multiple of SCAN_ITEMS): vector loads when
// all of them exist, one by one at the end of the table.  len[] receives the plain lengths.
template <bool FASTQ>
__device__ __forceinline__ void line_items(const uint32_t* __restrict__ line_len, const uint8_t* __restrict__ line_head,
                                           uint64_t base, uint64_t nlines, LineSum* v, uint32_t* len)
{
    static_assert(SCAN_ITEMS == 8, "two 16-byte loads of lengths, one 8-byte load of flags");
    if (base + SCAN_ITEMS <= nlines) {
        const uint4 l0 = *reinterpret_cast<const uint4*>(line_len + base), l1 = *reinterpret_cast<const uint4*>(line_len + base + 4);
        uint2 hd = make_uint2(0, 0);
        if (!FASTQ) hd = *reinterpret_cast<const uint2*>(line_head + base);
        const uint32_t l[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const bool head = FASTQ ? (k & 3) == 0 : (((k < 4 ? hd.x : hd.y) >> (8 * (k & 3))) & 0xffu) != 0;
            len[k] = l[k];
            v[k].heads = head ? 1u : 0u;
            if (FASTQ) v[k].bytes = (k & 3) == 1 ? l[k] : 0;       // base is a multiple of 4
            else v[k].bytes = head ? 0 : l[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) {
            v[k] = line_item<FASTQ>(line_len, line_head, base + k, nlines);
            len[k] = base + k < nlines ? line_len[base + k] : 0;
        }
    }
}

// Four-line FASTQ: the items of the scan are RECORDS (lines 4 r .. 4 r + 3, their lengths one 16-byte load):
// bytes = the length of the sequence line, heads = 1.  `base` is the thread's first record (SCAN_ITEMS consecutive
// ones are its items).  The loads are dealt the other way round -- the lanes of one load take consecutive records,
// 1 KiB per instruction, where a thread's own eight records lie 128 bytes from its neighbour's -- and the
// sequence lengths change hands through LDS.  With `check`, a quality line that is not as long as its sequence
// line is reported (complete records only).
__device__ __forceinline__ void record_items(const uint32_t* __restrict__ line_len, uint64_t base, uint64_t nlines, ParseState* st,
                                             bool check, LineSum* v, uint32_t* s_len)
{
    const uint64_t block0 = base - (uint64_t)threadIdx.x * SCAN_ITEMS;      // the workgroup's first record
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const uint64_t i0 = 4 * (block0 + (uint64_t)k * PB + threadIdx.x);
        uint32_t len = 0;
        if (i0 + 4 <= nlines) {
            const uint4 l = *reinterpret_cast<const uint4*>(line_len + i0);
            len = l.y;
            if (check && l.w != l.y) atomicMin(&st->err_line, (unsigned long long)(i0 + 3));
        } else if (i0 + 1 < nlines) {                     // the block ends inside the record
            len = line_len[i0 + 1];
        }
        s_len[k * PB + threadIdx.x] = len;
    }
    __syncthreads();
    const uint4 a = *reinterpret_cast<const uint4*>(s_len + threadIdx.x * SCAN_ITEMS);
    const uint4 b = *reinterpret_cast<const uint4*>(s_len + threadIdx.x * SCAN_ITEMS + 4);
    const uint32_t l[SCAN_ITEMS] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k].bytes = l[k];
        v[k].heads = 4 * (base + k) < nlines ? 1u : 0u;
    }
}

template <bool FASTQ>
__global__ __launch_bounds__(PB) void k_line_partials(const uint32_t* __restrict__ line_len,
                                                      const uint8_t* __restrict__ line_head, ParseState* st,
                                                      LineSum* __restrict__ part)
{
    __shared__ uint64_t s_b[PB / 64];
    __shared__ uint32_t s_h[PB / 64];
    const uint64_t nlines = st->nlines;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint64_t b = 0; uint32_t h = 0;
    LineSum v[SCAN_ITEMS];
    uint32_t len[SCAN_ITEMS];
    __shared__ __attribute__((aligned(16))) uint32_t s_len[FASTQ ? SCAN_TILE : 4];
    if (FASTQ) record_items(line_len, base, nlines, st, true, v, s_len);      // (a quality line must be as long as its sequence line)
    else line_items<FASTQ>(line_len, line_head, base, nlines, v, len);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { b += v[k].bytes; h += v[k].heads; }
    for (int d = 1; d < 64; d <<= 1) { b += __shfl_xor(b, d); h += __shfl_xor(h, d); }
    if ((threadIdx.x & 63) == 0) { s_b[threadIdx.x >> 6] = b; s_h[threadIdx.x >> 6] = h; }
    __syncthreads();
    if (threadIdx.x == 0) {
        LineSum t{0, 0};
        for (int w = 0; w < PB / 64; w++) { t.bytes += s_b[w]; t.heads += s_h[w]; }
        part[blockIdx.x] = t;
    }
}

// Exclusive scan of the workgroup sums, in place, by one workgroup: rounds of 4096 consecutive entries (four per
// thread), a shuffle scan per wavefront, the 16 wavefront totals through LDS, the running total carried in registers.
template <bool FASTQ>
__global__ __launch_bounds__(1024) void k_scan_partials(LineSum* __restrict__ part, uint64_t npart, ParseState* st, int final,
                                                        uint64_t max_records)
{
    constexpr int PER = 4;
    __shared__ uint64_t s_b[16];
    __shared__ uint32_t s_h[16];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t cb = 0; uint32_t ch = 0;                     // sum of everything before this round
    for (uint64_t base = 0; base < npart; base += 1024 * PER) {
        const uint64_t i = base + (uint64_t)PER * threadIdx.x;
        LineSum v[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) v[k] = i + k < npart ? part[i + k] : LineSum{0, 0};
        uint64_t mb = 0; uint32_t mh = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) { mb += v[k].bytes; mh += v[k].heads; }
        uint64_t ib = mb; uint32_t ih = mh;
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t tb = __shfl_up(ib, d); const uint32_t th = __shfl_up(ih, d);
            if ((int)lane >= d) { ib += tb; ih += th; }
        }
        if (lane == 63) { s_b[wave] = ib; s_h[wave] = ih; }
        __syncthreads();
        uint64_t rb = cb + ib - mb; uint32_t rh = ch + ih - mh;
#pragma unroll
        for (uint32_t w = 0; w < 16; w++) {
            const uint64_t tb = s_b[w]; const uint32_t th = s_h[w];
            if (w < wave) { rb += tb; rh += th; }
            cb += tb; ch += th;
        }
#pragma unroll
        for (int k = 0; k < PER; k++) {
            if (i + k < npart) { part[i + k].bytes = rb; part[i + k].heads = rh; }
            rb += v[k].bytes; rh += v[k].heads;
        }
        __syncthreads();                                  // the totals are rewritten in the next round
    }
    if (threadIdx.x == 0) {
        st->total.bytes = cb; st->total.heads = ch;
        // the records k_parse_decide will settle on, or more (it may find a format error and settle on none): FASTA
        // all head lines, but the last one's record when the text goes on; FASTQ the whole groups of four lines
        // before the first blank head line
        uint64_t cap;
        if (FASTQ) {
            const unsigned long long fbh = st->first_blank_head;
            const uint64_t leff = fbh != ~0ull && fbh < st->nlines ? fbh : st->nlines;
            cap = leff / 4;
        } else {
            cap = final ? ch : (ch > 1 ? ch - 1 : 0);
        }
        st->nrec_cap = cap <= max_records ? cap : 0;       // (more than the tables hold: a capacity error, nothing is written)
    }
}

// The scan proper, and what every head line contributes to the batch: its record's offset and name span (a head
// line of rank r opens record r; FASTA: a non-blank line before the first head line is a format error).  Records
// beyond `nrec_cap` -- the number of complete records, known before this pass from the head count or the line
// count, and never more than the tables hold -- are not written.
template <bool FASTQ, typename POS>
__global__ __launch_bounds__(PB) void k_line_scan(const uint32_t* __restrict__ line_len,
                                                  const uint8_t* __restrict__ line_head, ParseState* st,
                                                  const LineSum* __restrict__ part, POS* __restrict__ line_dst,
                                                  const POS* __restrict__ line_start,
                                                  uint64_t max_records, uint64_t* __restrict__ offsets,
                                                  uint64_t* __restrict__ name_pos, uint32_t* __restrict__ name_len)
{
    __shared__ uint64_t s_b[PB / 64];
    __shared__ uint32_t s_h[PB / 64];
    const uint64_t nlines = st->nlines;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    LineSum v[SCAN_ITEMS];
    uint32_t len[SCAN_ITEMS];
    uint64_t b = 0; uint32_t h = 0;
    __shared__ __attribute__((aligned(16))) uint32_t s_len[FASTQ ? SCAN_TILE : 4];
    if (FASTQ) record_items(line_len, base, nlines, st, false, v, s_len);
    else line_items<FASTQ>(line_len, line_head, base, nlines, v, len);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { b += v[k].bytes; h += v[k].heads; }
    // inclusive scan of the per-thread sums across the wavefront, then across wavefronts
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t ib = b; uint32_t ih = h;
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t tb = __shfl_up(ib, d); const uint32_t th = __shfl_up(ih, d);
        if ((int)lane >= d) { ib += tb; ih += th; }
    }
    if (lane == 63) { s_b[wave] = ib; s_h[wave] = ih; }
    __syncthreads();
    uint64_t rb = part[blockIdx.x].bytes + ib - b; uint32_t rh = part[blockIdx.x].heads + ih - h;
    for (uint32_t w = 0; w < wave; w++) { rb += s_b[w]; rh += s_h[w]; }
    const uint64_t nrec_cap = st->nrec_cap;
    const uint32_t nheads = st->total.heads;              // (from the scan of the workgroup sums)
    if (FASTQ) {
        // record base + k starts at the sum before it; one entry beyond the records the block can end up with, for
        // the decision to read (the names were written with the line table)
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) {
            if (v[k].heads && base + k <= nrec_cap && nrec_cap != 0) offsets[base + k] = rb;
            rb += v[k].bytes;
        }
        return;
    }
    if (base + SCAN_ITEMS <= nlines) {                    // 16-byte stores
        uint64_t d[SCAN_ITEMS]; uint32_t r[SCAN_ITEMS];
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) { d[k] = rb; r[k] = rh; rb += v[k].bytes; rh += v[k].heads; }
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) {
            if (v[k].heads) {
                if (r[k] < nrec_cap) { offsets[r[k]] = d[k]; name_pos[r[k]] = (uint64_t)line_start[base + k] + 1; name_len[r[k]] = len[k] - 1; }
                if (r[k] == 0) st->first_head = base + k;             // (one writer each)
                if (r[k] + 1 == nheads) st->last_head = base + k;
            } else if (!FASTQ && r[k] == 0 && len[k] != 0) {
                atomicMin(&st->err_line, (unsigned long long)(base + k));
            }
        }
        if constexpr (sizeof(POS) == 8) {
#pragma unroll
            for (int k = 0; k < SCAN_ITEMS; k += 2)
                *reinterpret_cast<uint4*>(line_dst + base + k) =
                    make_uint4((uint32_t)d[k], (uint32_t)(d[k] >> 32), (uint32_t)d[k + 1], (uint32_t)(d[k + 1] >> 32));
        } else {
#pragma unroll
            for (int k = 0; k < SCAN_ITEMS; k += 4)
                *reinterpret_cast<uint4*>(line_dst + base + k) =
                    make_uint4((uint32_t)d[k], (uint32_t)d[k + 1], (uint32_t)d[k + 2], (uint32_t)d[k + 3]);
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < nlines) {
            line_dst[base + k] = (POS)rb;
            if (v[k].heads) {
                if (rh < nrec_cap) { offsets[rh] = rb; name_pos[rh] = (uint64_t)line_start[base + k] + 1; name_len[rh] = len[k] - 1; }
                if (rh == 0) st->first_head = base + k;
                if (rh + 1 == nheads) st->last_head = base + k;
            } else if (!FASTQ && rh == 0 && len[k] != 0) {
                atomicMin(&st->err_line, (unsigned long long)(base + k));
            }
        }
        rb += v[k].bytes; rh += v[k].heads;
    }
}

// ---- pass 4: decide, emit, copy -----------------------------------------------------------

template <typename POS>
__device__ __forceinline__ uint64_t start_of_line(uint64_t k, uint64_t nlines, uint64_t n_nl, uint64_t n,
                                                  const POS* __restrict__ line_start,
                                                  const uint64_t* __restrict__ nlpos)
{
    if (k < nlines) return line_start[k];
    if (nlines > n_nl) return n;                         // the last line had no newline: nothing follows
    return nlines ? (nlpos[nlines - 1] & NL_POS) + 1 : 0;
}

// One thread: how many complete records the block holds and where the rest starts.
template <bool FASTQ, typename POS>
__global__ void k_parse_decide(ParseState* st, uint64_t n, int final, uint64_t max_lines,
                               const uint64_t* __restrict__ nlpos, const POS* __restrict__ line_start,
                               const POS* __restrict__ line_dst,
                               uint64_t max_records, uint64_t* __restrict__ offsets, uint64_t seq_capacity,
                               ParseResult* __restrict__ res)
{
    const uint64_t L = st->nlines, n_nl = st->n_nl;
    uint64_t nrec = 0, seq_bytes = 0, consumed = 0;
    uint32_t status = PARSE_OK;
    if (n_nl + 1 > max_lines) {
        status = PARSE_ECAPACITY;                        // the line tables were too small for this text
    } else if (!FASTQ) {
        const uint64_t H = st->total.heads;
        st->copy_lo = st->first_head;                     // (no head line: ~0, no line is copied)
        if (final) {
            nrec = H; seq_bytes = st->total.bytes; consumed = n;
            st->copy_hi = L;
        } else if (H > 1) {
            // the record of the last head line may continue in the next block: stop before it
            const uint64_t last_head = st->last_head;
            nrec = H - 1; seq_bytes = line_dst[last_head]; consumed = line_start[last_head];
            st->copy_hi = last_head;
        }
    } else {
        const unsigned long long err = st->err_line, fbh = st->first_blank_head;
        uint64_t Leff = L;
        bool bad = false; uint64_t bad_line = 0;
        if (fbh != ~0ull && fbh < Leff) {
            Leff = fbh;                                   // records end at the first blank head line ...
            if (final && st->any_nonblank && st->last_nonblank > fbh) { bad = true; bad_line = fbh; }   // ... only blank lines may follow
        }
        nrec = Leff / 4;
        if (final && (Leff & 3) != 0 && !bad) { bad = true; bad_line = nrec * 4; }    // truncated last record
        if (err != ~0ull && (err < nrec * 4 || final) && !(bad && bad_line < err)) { bad = true; bad_line = err; }
        if (bad) { status = PARSE_EFORMAT; st->err_line = bad_line; nrec = 0; }
        else {
            st->err_line = ~0ull;
            // where record nrec would start: the scan left it in offsets[nrec] (nothing was written for a block with
            // more records than the tables hold)
            const uint64_t R = (L + 3) / 4;
            seq_bytes = (nrec == 0 || nrec > max_records) ? 0 : (nrec < R ? offsets[nrec] : st->total.bytes);
            consumed = final ? n : start_of_line(nrec * 4, L, n_nl, n, line_start, nlpos);
        }
    }
    st->status = status; st->nrec = nrec; st->seq_bytes = seq_bytes; st->consumed = consumed;
    if (status == PARSE_OK && nrec <= max_records) offsets[nrec] = seq_bytes;
    // what the caller reads (the copies that follow change nothing of it)
    ParseResult r;
    r.nrec = nrec; r.seq_bytes = seq_bytes; r.consumed = consumed; r.lines = L;
    r.status = status; r.pad = 0; r.error_line = -1;
    const unsigned long long err_line = st->err_line;
    if (r.status == PARSE_OK && err_line != ~0ull) r.status = PARSE_EFORMAT;
    if (r.status == PARSE_EFORMAT) { r.error_line = (int64_t)err_line; r.nrec = 0; r.seq_bytes = 0; r.consumed = 0; }
    if (r.status == PARSE_OK && (r.nrec > max_records || r.seq_bytes > seq_capacity)) r.status = PARSE_ECAPACITY;
    *res = r;
}

__device__ __forceinline__ uint64_t shfl64(uint64_t v, uint32_t src_lane)
{
    return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)src_lane) << 32) |
           (uint32_t)__shfl((int)(uint32_t)v, (int)src_lane);
}

// One line of at least WIDE_LINE bytes, by all 64 lanes of a wavefront: the stores are aligned 16-byte chunks of
// the plane (the bytes before the first whole chunk go one per lane, the end of the line as its last 16 bytes
// once more), the loads take any alignment; the loads of four rounds of chunks go out before the first store.
constexpr uint32_t WIDE_LINE = 512;

__device__ __forceinline__ void copy_line_wave(uint8_t* __restrict__ out, const uint8_t* __restrict__ src, uint32_t len,
                                               uint32_t lane)
{
    const uint32_t head = (uint32_t)(-reinterpret_cast<uintptr_t>(out)) & 15u;       // len >= WIDE_LINE > head + 16
    if (lane < head) out[lane] = src[lane];
    constexpr int PU = 4;
    for (uint32_t p0 = head + lane * 16; p0 < len; p0 += 64 * 16 * PU) {
        uint4 v[PU];
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const uint32_t p = p0 + u * 1024;
            v[u] = p < len ? ldu16(src + (p + 16 <= len ? p : len - 16)) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const uint32_t p = p0 + u * 1024;
            if (p < len) { const uint4 t = v[u]; __builtin_memcpy(out + (p + 16 <= len ? p : len - 16), &t, 16); }
        }
    }
}

// One line's bytes, by the G lanes of its group (lane j): 16-byte pieces at any alignment, the end of the
// line as its last 16 bytes copied once more (same values, no byte loop).
template <uint32_t G>
__device__ __forceinline__ void copy_line(uint8_t* __restrict__ out, const uint8_t* __restrict__ src, uint32_t len, uint32_t j)
{
    for (uint32_t p = j * 16; p < len; p += G * 16) {
        uint4 v;
        if (p + 16 <= len) {
            __builtin_memcpy(&v, src + p, 16);
            __builtin_memcpy(out + p, &v, 16);
        } else if (len >= 16) {
            __builtin_memcpy(&v, src + len - 16, 16);
            __builtin_memcpy(out + len - 16, &v, 16);
        } else {
            for (uint32_t k = p; k < len; k++) out[k] = src[k];
        }
    }
}

// Sequence (and quality) lines to their places in the planes.  The lines a wavefront takes in one step
// are consecutive, so what they write is one contiguous run of each plane: as many lines as fit a stage
// of PC_STAGE bytes per plane are put together in LDS (8 lanes per line; a line lands at any byte there at
// no cost) and leave as aligned 16-byte chunks in whole-wavefront rows, the bytes that share a chunk with
// the neighbouring step byte by byte.  FASTQ: only the second and fourth line of a record are copied, so
// the work items are those (two per record); FASTA: all lines (head lines contribute nothing).  A step
// whose run does not fit the stage is copied straight to global memory; lines beyond LONG_LINE are left
// to k_parse_copy_long.
constexpr uint32_t PC_STAGE = 3584;

template <bool FASTQ, typename POS>
__attribute__((amdgpu_waves_per_eu(FASTQ ? 5 : 8, FASTQ ? 5 : 8)))
__global__ __launch_bounds__(PB) void k_parse_copy(const ParseState* st, uint64_t max_records, uint64_t seq_capacity,
                                                   const uint8_t* __restrict__ text,
                                                   const POS* __restrict__ line_start,
                                                   const uint32_t* __restrict__ line_len,
                                                   const uint8_t* __restrict__ line_head,
                                                   const POS* __restrict__ line_dst,
                                                   const uint64_t* __restrict__ offsets,
                                                   uint8_t* __restrict__ seq, uint8_t* __restrict__ qual)
{
    constexpr uint32_t G = 8, GROUPS = 64 / G, NPLANE = FASTQ ? 2 : 1;
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[PB / 64][NPLANE][PC_STAGE + 32];
    const uint32_t lane = threadIdx.x & 63, j = lane & (G - 1), grp = lane / G;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t nrec = st->nrec, nlines = st->nlines, seq_bytes = st->seq_bytes;
    const uint64_t copy_lo = st->copy_lo, copy_hi = st->copy_hi;
    if (st->status != PARSE_OK || nrec > max_records || seq_bytes > seq_capacity) return;
    const uint64_t nwork = FASTQ ? 2 * ((nlines + 3) / 4) : nlines;
    // work items per step, from the mean number of bytes an item brings
    uint32_t W;
    {
        const uint64_t units = FASTQ ? nrec : nlines;
        const uint64_t mean = units ? seq_bytes / units + 1 : 1;
        uint64_t k = PC_STAGE * 7 / 8 / mean;
        const uint64_t kmax = FASTQ ? 32 : 64;
        k = k < 1 ? 1 : (k > kmax ? kmax : k);
        W = (uint32_t)(FASTQ ? 2 * k : k);
    }
    const uint64_t nsteps = (nwork + W - 1) / W;
    for (uint64_t step = (uint64_t)blockIdx.x * (PB / 64) + wave; step < nsteps; step += (uint64_t)gridDim.x * (PB / 64)) {
        // lane i holds the tables of work item step * W + i
        const uint64_t w = step * W + lane;
        uint64_t t_dst = 0, t_start = 0;
        uint32_t t_len = 0, t_q = 0;
        bool valid = false;
        if (lane < W && w < nwork) {
            const uint64_t i = FASTQ ? 4 * (w >> 1) + 1 + 2 * (w & 1) : w;
            if (i < nlines) {
                if (FASTQ) {
                    const uint32_t role = (uint32_t)(i & 3);
                    if ((i >> 2) < nrec) { valid = true; t_dst = offsets[i >> 2]; t_q = role == 3; }
                } else {
                    if (i > copy_lo && i < copy_hi && !line_head[i]) { valid = true; t_dst = line_dst[i]; }
                }
                if (valid) { t_len = line_len[i]; t_start = line_start[i]; }
            }
        }
        const unsigned long long m = __ballot(valid);
        if (m == 0) continue;
        const uint32_t first = (uint32_t)__builtin_ctzll(m), last = 63u - (uint32_t)__builtin_clzll(m);
        const uint64_t t_end = t_dst + t_len;
        const uint64_t p0 = shfl64(t_dst, first);
        uint64_t p1 = shfl64(t_end, last);
        if (FASTQ && last > 0) { const uint64_t e2 = shfl64(t_end, last - 1); p1 = e2 > p1 ? e2 : p1; }
        const uint64_t a0 = p0 & ~15ull;
        const bool staged = !__any(valid && t_len > LONG_LINE) && p1 - a0 <= PC_STAGE;
        if (!staged) {
            for (uint32_t base = 0; base <= last; base += GROUPS) {   // every lane takes part in the cross-lane reads
                const uint32_t i = base + grp;
                const uint32_t len = (uint32_t)__shfl((int)t_len, (int)i), q = (uint32_t)__shfl((int)t_q, (int)i);
                const uint64_t dst = shfl64(t_dst, i), start = shfl64(t_start, i);
                if (len == 0 || len >= WIDE_LINE) continue;            // (an item that is not valid has len 0)
                copy_line<G>((FASTQ && q ? qual : seq) + dst, text + start, len, j);
            }
            // lines of half a kilobyte and more, one after the other by the whole wavefront
            unsigned long long wide = __ballot(valid && t_len >= WIDE_LINE && t_len <= LONG_LINE);
            while (wide) {
                const uint32_t i = (uint32_t)__builtin_ctzll(wide);
                wide &= wide - 1;
                const uint32_t len = (uint32_t)__shfl((int)t_len, (int)i), q = (uint32_t)__shfl((int)t_q, (int)i);
                copy_line_wave((FASTQ && q ? qual : seq) + shfl64(t_dst, i), text + shfl64(t_start, i), len, lane);
            }
            continue;
        }
        // The 16-byte pieces of all lines of the step dealt over the 64 lanes: a line gets PF piece slots (the
        // most any line of the step needs), slot q = lane + 64 k is piece q mod PF of line q / PF -- lanes are busy
        // whatever the line length, where 8 lanes per line leave six of sixteen slots empty on a 150-base read.
        // (FASTQ: the bases and the qualities of a record are neighbours in the text and stay neighbours in an
        // instruction; which stage a piece goes to is the parity of its line.)
        const uint32_t d_rel = (uint32_t)(t_dst - a0);              // (0-length items never use theirs)
        const uint64_t start0 = shfl64(t_start, first);
        const uint32_t st_rel = (uint32_t)(t_start - start0);        // the lines of a step lie within 4 GiB of text
        uint32_t PF = (t_len + 15) >> 4;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)PF, d); PF = t > PF ? t : PF; }
        if (PF != 0) {
            const uint32_t nslots = (last + 1) * PF;
            const uint32_t magic = PF > 1 ? (uint32_t)((0x100000000ull + PF - 1) / PF) : 0u;   // q / PF for q < 2^16
            // PU rounds of slots at a time, the loads of all of them ahead of the first LDS write
            constexpr int PU = 4;                                     // (2 and 7 measure the same)
            uint8_t* const stage0 = s_stage[wave][0];
            for (uint32_t q0 = 0; q0 < nslots; q0 += 64 * PU) {       // every lane takes part in the cross-lane reads
                uint4 v[PU];
                uint32_t at[PU];
                bool act[PU];
#pragma unroll
                for (int u = 0; u < PU; u++) {
                    const uint32_t qq = q0 + 64 * u + lane;
                    const uint32_t q = qq < nslots ? qq : nslots - 1;
                    const uint32_t r = PF > 1 ? __umulhi(q, magic) : q, k = q - r * PF;
                    const uint32_t len = (uint32_t)__shfl((int)t_len, (int)r), dr = (uint32_t)__shfl((int)d_rel, (int)r);
                    const uint32_t sr = (uint32_t)__shfl((int)st_rel, (int)r);
                    const uint32_t p = 16 * k;
                    const bool in = qq < nslots && p < len;
                    const uint32_t plane = FASTQ ? (r & 1u) * (PC_STAGE + 32) : 0u;
                    act[u] = in && len >= 16;
                    const uint32_t off = p + 16 <= len ? p : len - 16;   // a line's last piece: its last 16 bytes again
                    at[u] = plane + dr + off;
                    const uint8_t* src = text + start0 + sr;
                    v[u] = act[u] ? ldu16(src + off) : make_uint4(0, 0, 0, 0);
                    if (in && len < 16)
                        for (uint32_t b = 0; b < len; b++) stage0[plane + dr + b] = src[b];
                }
#pragma unroll
                for (int u = 0; u < PU; u++)
                    if (act[u]) { const uint4 t = v[u]; __builtin_memcpy(stage0 + at[u], &t, 16); }
            }
        }
        wave_lds_handover();
        const uint32_t nbytes = (uint32_t)(p1 - a0), head = (uint32_t)(p0 - a0);
        const uint32_t c_lo = head ? 16u : 0u, c_hi = nbytes & ~15u;
#pragma unroll
        for (uint32_t pl = 0; pl < NPLANE; pl++) {
            uint8_t* out = pl ? qual : seq;
            const uint8_t* stage = s_stage[wave][pl];
            for (uint32_t c = lane * 16; c < c_hi; c += 1024)
                if (c >= c_lo) *reinterpret_cast<uint4*>(out + a0 + c) = *reinterpret_cast<const uint4*>(stage + c);
            if (head && lane >= head && lane < 16 && lane < nbytes) out[a0 + lane] = stage[lane];
            if (c_hi < nbytes && (c_hi != 0 || !head) && lane < nbytes - c_hi) out[a0 + c_hi + lane] = stage[c_hi + lane];
        }
        wave_lds_handover();
    }
}

// The lines longer than LONG_LINE (unwrapped chromosome-length sequences): workgroups own
// aligned windows of the text, and only the lines that hold a window's first and last byte can be
// that long, so such a line is copied by as many workgroups as it has windows.
template <bool FASTQ, typename POS>
__global__ __launch_bounds__(PB) void k_parse_copy_long(const ParseState* st, uint64_t max_records, uint64_t seq_capacity,
                                                        const uint8_t* __restrict__ text, uint64_t n,
                                                        const POS* __restrict__ line_start,
                                                        const uint32_t* __restrict__ line_len,
                                                        const uint8_t* __restrict__ line_head,
                                                        const POS* __restrict__ line_dst,
                                                        const uint64_t* __restrict__ offsets,
                                                        uint8_t* __restrict__ seq, uint8_t* __restrict__ qual)
{
    const uint64_t nrec = st->nrec, nlines = st->nlines;
    if (st->n_long == 0 || nlines == 0) return;
    if (st->status != PARSE_OK || nrec > max_records || st->seq_bytes > seq_capacity) return;
    const uint64_t ntiles = (n + LONG_LINE - 1) / LONG_LINE;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t t0 = tile * LONG_LINE, t1 = t0 + LONG_LINE < n ? t0 + LONG_LINE : n;
        uint64_t which[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {                    // last line that starts at or before the byte
            const uint64_t p = c == 0 ? t0 : t1 - 1;
            uint64_t lo = 0, hi = nlines;                // line_start[lo] <= p (line 0 starts at 0)
            while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (line_start[mid] <= p) lo = mid; else hi = mid; }
            which[c] = lo;
        }
#pragma unroll 1
        for (int c = 0; c < 2; c++) {
            if (c == 1 && which[1] == which[0]) break;
            const uint64_t i = which[c];
            const uint64_t len = line_len[i];
            if (len <= LONG_LINE) continue;              // block-uniform
            uint8_t* dst_plane;
            uint64_t dst;
            if (FASTQ) {
                const uint32_t role = (uint32_t)(i & 3);
                if ((role != 1 && role != 3) || (i >> 2) >= nrec) continue;
                dst_plane = role == 1 ? seq : qual;
                dst = offsets[i >> 2];
            } else {
                if (!(i > st->copy_lo && i < st->copy_hi) || line_head[i]) continue;
                dst_plane = seq;
                dst = line_dst[i];
            }
            const uint64_t ls = line_start[i];
            const uint64_t k0 = (ls > t0 ? ls : t0) - ls, k1 = (ls + len < t1 ? ls + len : t1) - ls;
            const uint8_t* src = text + ls;
            uint8_t* out = dst_plane + dst;
            for (uint64_t p = k0 + 16ull * threadIdx.x; p < k1; p += 16ull * PB) {
                if (p + 16 <= k1) {
                    uint4 v;
                    __builtin_memcpy(&v, src + p, 16);
                    __builtin_memcpy(out + p, &v, 16);
                } else {
                    for (uint64_t k = p; k < k1; k++) out[k] = src[k];
                }
            }
        }
    }
}


// ---- FASTQ whose records are wrapped over several lines -----------------------------------------------------
//
// The reader the reference uses accepts FASTQ records with sequence and quality on several lines (README.md:88,
// doc/docs/usage.md:357-359 of the reference): a record is a line that starts with '@', sequence lines up to the
// first line that starts with '+', and quality lines until they hold as many bytes as the sequence; blank lines
// may stand between records.  '@' and '+' may start a quality line, so which lines start records depends on every
// record before -- but where a record WOULD end if it started at a given '@' line depends on that line alone: the
// next '+' line, the bytes between, and the first line at which the bytes after the '+' line reach that number
// (prefix sums of the line lengths, a binary search).  Every '@' line thus gets the line its successor would start
// at, and the records of the text are the chain of successors from the first non-blank line: marked by pointer
// doubling (a marked line marks its successor, successors are squared; log2(lines) rounds).  Anything else --
// a chain that reaches a line that is not an '@' line, quality bytes that overshoot, an incomplete record at the
// end of a file -- is reported as a format error without a line number: the caller's serial scan names the line.
// All passes are element-wise over the line table or sums over it; nothing here is tuned (such files are rare,
// and the serial rewrite they replace runs at 1 GB/s).
struct WrapState { uint32_t status, wrapped; uint64_t consumed, lines, nrec; };
constexpr uint32_t WQ_B = 256;

// tmp[i] for the sums over lines: kind 0 = length, 1 = is a '+' line, 2 = is not blank (zero from line L on)
__global__ __launch_bounds__(WQ_B) void k_wq_flags(const ParseState* st, const uint32_t* __restrict__ line_len,
                                                   const uint8_t* __restrict__ line_first, int kind, uint64_t cap,
                                                   uint64_t* __restrict__ tmp)
{
    const uint64_t L = st->nlines;
    for (uint64_t i = (uint64_t)blockIdx.x * WQ_B + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * WQ_B) {
        uint64_t v = 0;
        if (i < L) {
            const uint32_t len = line_len[i];
            v = kind == 0 ? len : (kind == 1 ? (len && line_first[i] == '+') : (len != 0));
        }
        tmp[i] = v;
    }
}

// the c-th '+' line and the c-th non-blank line (entry number-of-them = L): first such line at or after i is
// idx[count before i]
__global__ __launch_bounds__(WQ_B) void k_wq_index(const ParseState* st, const uint32_t* __restrict__ line_len,
                                                   const uint8_t* __restrict__ line_first, const uint64_t* __restrict__ cPlus,
                                                   const uint64_t* __restrict__ cNb, uint64_t cap,
                                                   uint32_t* __restrict__ plus_idx, uint32_t* __restrict__ nb_idx)
{
    const uint64_t L = st->nlines;
    for (uint64_t i = (uint64_t)blockIdx.x * WQ_B + threadIdx.x; i <= L && i < cap; i += (uint64_t)gridDim.x * WQ_B) {
        if (i == L) { plus_idx[cPlus[L]] = (uint32_t)L; nb_idx[cNb[L]] = (uint32_t)L; continue; }
        const uint32_t len = line_len[i];
        if (len) {
            nb_idx[cNb[i]] = (uint32_t)i;
            if (line_first[i] == '+') plus_idx[cPlus[i]] = (uint32_t)i;
        }
    }
}

// The record that would start at line i: its '+' line and its last line; false: there is none (*why: L + 1 =
// the text ends before the record does, L + 2 = not a record).
__device__ __forceinline__ bool wq_record(uint64_t i, uint64_t L, int final, const uint32_t* __restrict__ line_len,
                                          const uint8_t* __restrict__ line_first, const uint64_t* __restrict__ cP,
                                          const uint64_t* __restrict__ cPlus, const uint32_t* __restrict__ plus_idx,
                                          uint64_t& plus, uint64_t& last, uint32_t& why)
{
    const uint32_t cut = (uint32_t)(final ? L + 2 : L + 1);
    if (line_len[i] == 0 || line_first[i] != '@') { why = (uint32_t)(L + 2); return false; }
    const uint64_t j = plus_idx[cPlus[i + 1]];
    if (j >= L) {
        // no '+' line any more: where the file ends, a head with nothing but blank lines behind it is a record
        // without bases (the reader restated from the reference takes it so); bases without quality are not
        if (final && cP[L] == cP[i + 1]) { plus = L; last = L - 1; return true; }
        why = cut; return false;
    }
    const uint64_t S = cP[j] - cP[i + 1];
    plus = j; last = j;
    if (S == 0) return true;
    const uint64_t q0 = cP[j + 1];
    if (cP[L] - q0 < S) { why = cut; return false; }
    // smallest k with cP[k + 1] - q0 >= S: the quality is as many lines as the sequence or so, so the search gallops
    // from the '+' line (1, 2, 4, ... lines on) before it halves what is left
    uint64_t lo = j + 1, hi = L - 1, step = 1;
    while (lo + step - 1 < hi && cP[lo + step] - q0 < S) { lo += step; step <<= 1; }        // (k = lo + step - 1 is too small)
    if (lo + step - 1 < hi) hi = lo + step - 1;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (cP[mid + 1] - q0 >= S) hi = mid; else lo = mid + 1; }
    if (cP[lo + 1] - q0 != S) { why = (uint32_t)(L + 2); return false; }       // more quality than sequence
    last = lo;
    return true;
}

// successor of every line; three lines behind the text stand for "the end", "cut" and "not a record"
__global__ __launch_bounds__(WQ_B) void k_wq_next(const ParseState* st, int final, const uint32_t* __restrict__ line_len,
                                                  const uint8_t* __restrict__ line_first, const uint64_t* __restrict__ cP,
                                                  const uint64_t* __restrict__ cPlus, const uint64_t* __restrict__ cNb,
                                                  const uint32_t* __restrict__ plus_idx, const uint32_t* __restrict__ nb_idx,
                                                  uint64_t cap, uint32_t* __restrict__ nxt0, uint32_t* __restrict__ nxt,
                                                  uint8_t* __restrict__ mark)
{
    const uint64_t L = st->nlines;
    for (uint64_t i = (uint64_t)blockIdx.x * WQ_B + threadIdx.x; i < L + 3 && i < cap; i += (uint64_t)gridDim.x * WQ_B) {
        uint32_t to = (uint32_t)i;                        // (the three lines behind the text lead to themselves)
        if (i < L) {
            uint64_t plus, last; uint32_t why = 0;
            if (wq_record(i, L, final, line_len, line_first, cP, cPlus, plus_idx, plus, last, why)) to = nb_idx[cNb[last + 1]];
            else to = why;
        }
        nxt0[i] = to; nxt[i] = to;
        mark[i] = i == nb_idx[0] ? 1 : 0;                 // the first line that is not blank (L: there is none)
    }
}

__global__ __launch_bounds__(WQ_B) void k_wq_jump(const ParseState* st, uint64_t cap, const uint32_t* __restrict__ a,
                                                  uint32_t* __restrict__ b, uint8_t* __restrict__ mark)
{
    const uint64_t L = st->nlines;
    for (uint64_t i = (uint64_t)blockIdx.x * WQ_B + threadIdx.x; i < L + 3 && i < cap; i += (uint64_t)gridDim.x * WQ_B) {
        const uint32_t to = a[i];
        if (mark[i]) mark[to] = 1;
        b[i] = a[to];
    }
}

// the marked lines whose record is complete start the records of the block; where the chain ends
__global__ __launch_bounds__(WQ_B) void k_wq_emit(const ParseState* st, uint64_t n, int final, const uint64_t* __restrict__ nlpos,
                                                  const uint64_t* __restrict__ line_start, const uint32_t* __restrict__ nxt0,
                                                  const uint8_t* __restrict__ mark, uint64_t cap, uint64_t* __restrict__ tmp,
                                                  WrapState* w)
{
    const uint64_t L = st->nlines;
    for (uint64_t i = (uint64_t)blockIdx.x * WQ_B + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * WQ_B) {
        uint64_t e = 0;
        if (i < L && mark[i]) {
            if (nxt0[i] <= L) e = 1;
            else if (nxt0[i] == L + 1) { w->consumed = line_start[i]; w->lines = i; }
        }
        tmp[i] = e;
        if (i == L) {
            if (mark[L + 2]) w->status = PARSE_EFORMAT;
            if (mark[L]) {
                w->lines = L;
                w->consumed = final ? n : (L ? (nlpos[L - 1] & NL_POS) + 1 : 0);
            }
        }
    }
}

__global__ __launch_bounds__(WQ_B) void k_wq_records(const ParseState* st, int final, const uint32_t* __restrict__ line_len,
                                                     const uint8_t* __restrict__ line_first, const uint64_t* __restrict__ line_start,
                                                     const uint64_t* __restrict__ cP, const uint64_t* __restrict__ cPlus,
                                                     const uint64_t* __restrict__ cNb, const uint32_t* __restrict__ plus_idx,
                                                     const uint32_t* __restrict__ nb_idx, const uint64_t* __restrict__ cRid,
                                                     const uint64_t* __restrict__ is_start, uint64_t cap, uint64_t max_records,
                                                     uint32_t* __restrict__ rec_start, uint32_t* __restrict__ rec_plus,
                                                     uint32_t* __restrict__ rec_last, uint64_t* __restrict__ rec_S,
                                                     uint64_t* __restrict__ name_pos, uint32_t* __restrict__ name_len, WrapState* w)
{
    const uint64_t L = st->nlines;
    for (uint64_t i = (uint64_t)blockIdx.x * WQ_B + threadIdx.x; i < L && i < cap; i += (uint64_t)gridDim.x * WQ_B) {
        if (!is_start[i]) continue;
        const uint64_t r = cRid[i];
        if (r >= max_records) continue;
        uint64_t plus = 0, last = 0; uint32_t why;
        wq_record(i, L, final, line_len, line_first, cP, cPlus, plus_idx, plus, last, why);
        rec_start[r] = (uint32_t)i; rec_plus[r] = (uint32_t)plus; rec_last[r] = (uint32_t)last;
        rec_S[r] = cP[plus] - cP[i + 1];
        name_pos[r] = line_start[i] + 1; name_len[r] = line_len[i] - 1;
        // anything but '@', one sequence line, '+', one quality line, the next record: the caller may want to know
        if (plus != i + 2 || last != plus + 1 || nb_idx[cNb[last + 1]] != last + 1 || (r == 0 && i != 0)) w->wrapped = 1;
    }
}

__global__ void k_wq_check(const ParseState* st, uint64_t max_lines, const uint64_t* __restrict__ nrec_total,
                           const uint64_t* __restrict__ offsets, uint64_t max_records, uint64_t seq_capacity, WrapState* w)
{
    const uint64_t nrec = *nrec_total;
    w->nrec = nrec;
    if (st->n_nl + 1 > max_lines) w->status = PARSE_ECAPACITY;         // the line tables were too small for this text
    if (w->status == PARSE_OK && (nrec > max_records || offsets[nrec] > seq_capacity)) w->status = PARSE_ECAPACITY;
}

// every sequence and quality line to its place: a lane looks up where its line goes (64 lines per wavefront at a
// time, their lookups in flight together), then 8 lanes copy a line
__global__ __launch_bounds__(WQ_B) void k_wq_copy(const ParseState* st, const WrapState* w, const uint8_t* __restrict__ text,
                                                  const uint64_t* __restrict__ line_start, const uint32_t* __restrict__ line_len,
                                                  const uint64_t* __restrict__ cP, const uint64_t* __restrict__ cRid,
                                                  const uint32_t* __restrict__ rec_start, const uint32_t* __restrict__ rec_plus,
                                                  const uint32_t* __restrict__ rec_last, const uint64_t* __restrict__ offsets,
                                                  uint64_t cap, uint8_t* __restrict__ seq, uint8_t* __restrict__ qual)
{
    if (w->status != PARSE_OK) return;
    const uint64_t L = st->nlines;
    const uint32_t lane = threadIdx.x & 63, j = lane & 7, grp = lane >> 3;
    const uint64_t wave0 = ((uint64_t)blockIdx.x * WQ_B + threadIdx.x) >> 6, nwave = ((uint64_t)gridDim.x * WQ_B) >> 6;
    for (uint64_t base = wave0 * 64; base < L && base < cap; base += nwave * 64) {      // wave-uniform
        const uint64_t i = base + lane;
        uint64_t src = 0, dst = 0;
        uint32_t len = 0, is_q = 0;
        if (i < L && i < cap) {
            const uint64_t before = cRid[i + 1];          // records that start at or before line i
            const uint32_t l = line_len[i];
            if (before != 0 && l != 0) {
                const uint64_t r = before - 1;
                const uint64_t start = rec_start[r], plus = rec_plus[r], last = rec_last[r];
                if (i != start && i != plus && i <= last) {
                    is_q = i > plus ? 1u : 0u;
                    dst = offsets[r] + (cP[i] - cP[(is_q ? plus : start) + 1]);
                    src = line_start[i];
                    len = l;
                }
            }
        }
#pragma unroll 1
        for (uint32_t r8 = 0; r8 < 8; r8++) {
            const uint32_t from = r8 * 8 + grp;
            const uint32_t ll = (uint32_t)__shfl((int)len, (int)from), qq = (uint32_t)__shfl((int)is_q, (int)from);
            const uint64_t ss = shfl64(src, from), dd = shfl64(dst, from);
            if (ll) copy_line<8>((qq ? qual : seq) + dd, text + ss, ll, j);
        }
    }
}

__global__ void k_wq_finish(const WrapState* w, const uint64_t* __restrict__ offsets, ParseResult* __restrict__ res)
{
    ParseResult r;
    r.status = w->status; r.error_line = -1; r.pad = w->wrapped;
    r.nrec = 0; r.seq_bytes = 0; r.consumed = 0; r.lines = w->lines;
    if (r.status == PARSE_OK) { r.nrec = w->nrec; r.seq_bytes = offsets[w->nrec]; r.consumed = w->consumed; }
    *res = r;
}

}  // namespace

// ---- launcher ------------------------------------------------------------------------------

static inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

// Workspace layout for `max_lines` lines and `text_bytes` of text (all sections 256-byte aligned).
struct ParseLayout {
    uint64_t state, tile_count, tile_base, tile_last, tile_part, nl_tile, nlpos, line_start, line_len, line_head, line_dst, part, end;
    ParseLayout(uint64_t text_bytes, uint64_t max_lines)
    {
        const uint64_t ntile = (text_bytes + TILE - 1) / TILE + 1;
        const uint64_t npart = (max_lines + SCAN_TILE - 1) / SCAN_TILE + 1;
        uint64_t o = 0;
        state = o; o = align_up(o + sizeof(ParseState), 256);
        tile_count = o; o = align_up(o + ntile * 4, 256);
        tile_base = o; o = align_up(o + (ntile + 1) * 8, 256);
        tile_last = o; o = align_up(o + ntile * 8, 256);
        tile_part = o; o = align_up(o + (ntile / TS_TILE + 2) * 8, 256);
        nl_tile = o; o = align_up(o + ntile * (uint64_t)TILE * 4, 256);      // address space: only the entries that exist are touched
        nlpos = o; o = align_up(o + max_lines * 8, 256);
        line_start = o; o = align_up(o + max_lines * 8, 256);
        line_len = o; o = align_up(o + max_lines * 4, 256);
        line_head = o; o = align_up(o + max_lines, 256);
        line_dst = o; o = align_up(o + max_lines * 8, 256);
        part = o; o = align_up(o + npart * sizeof(LineSum), 256);
        end = o;
    }
};

uint64_t parse_workspace_size(uint64_t text_bytes, uint64_t max_lines)
{
    return ParseLayout(text_bytes, max_lines).end;
}

template <bool FASTQ, typename POS>
static hipError_t launch_parse_t(const uint8_t* text, uint64_t n, int final, uint8_t* seq, uint8_t* qual,
                                 uint64_t seq_capacity, uint64_t* offsets, uint64_t* name_pos, uint32_t* name_len,
                                 uint64_t max_records, uint8_t* ws, uint64_t max_lines, ParseResult* res, hipStream_t s)
{
    const ParseLayout lay(n, max_lines);
    ParseState* st = reinterpret_cast<ParseState*>(ws + lay.state);
    uint32_t* tile_count = reinterpret_cast<uint32_t*>(ws + lay.tile_count);
    uint64_t* tile_base = reinterpret_cast<uint64_t*>(ws + lay.tile_base);
    uint64_t* tile_last = reinterpret_cast<uint64_t*>(ws + lay.tile_last);
    uint32_t* nl_tile = reinterpret_cast<uint32_t*>(ws + lay.nl_tile);
    uint64_t* nlpos = reinterpret_cast<uint64_t*>(ws + lay.nlpos);
    POS* line_start = reinterpret_cast<POS*>(ws + lay.line_start);     // (the sections are sized for 64-bit entries)
    uint32_t* line_len = reinterpret_cast<uint32_t*>(ws + lay.line_len);
    uint8_t* line_head = ws + lay.line_head;
    POS* line_dst = reinterpret_cast<POS*>(ws + lay.line_dst);
    LineSum* part = reinterpret_cast<LineSum*>(ws + lay.part);

    const uint64_t ntile = (n + TILE - 1) / TILE;
    const uint32_t gt = (uint32_t)(ntile ? ntile : 1);
    // the line passes are sized for the most lines the tables can hold (the true count is
    // only known on the device); surplus threads leave at once
    const uint64_t lines_cap = max_lines < n + 1 ? max_lines : n + 1;
    auto capped = [](uint64_t blocks) { return (uint32_t)(blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks)); };
    const uint32_t gl = capped((lines_cap + PB - 1) / PB);            // grid-stride kernels
    // the items of the scan: lines, for four-line FASTQ records
    const uint64_t items_cap = FASTQ ? (lines_cap + 3) / 4 : lines_cap;
    const uint64_t npart = (items_cap + SCAN_TILE - 1) / SCAN_TILE ? (items_cap + SCAN_TILE - 1) / SCAN_TILE : 1;
    const uint32_t gc = capped((lines_cap * 8 + PB - 1) / PB);

    hipLaunchKernelGGL(k_nl_index, dim3(gt), dim3(PB), 0, s, text, n, tile_count, tile_last, nl_tile);
    launch_tile_scan(tile_count, ntile, tile_base, reinterpret_cast<uint64_t*>(ws + lay.tile_part), st, n, final, text, max_lines, s);
    hipLaunchKernelGGL((k_line_table<FASTQ, POS>), dim3(gl < 2048 ? gl : 2048), dim3(PB), 0, s, text, n, ntile, nl_tile, tile_count, tile_base,
                       tile_last, st, nlpos, line_start, line_len, line_head, 0, max_records, name_pos, name_len);
    hipLaunchKernelGGL((k_line_partials<FASTQ>), dim3((uint32_t)npart), dim3(PB), 0, s, line_len, line_head, st, part);
    hipLaunchKernelGGL((k_scan_partials<FASTQ>), dim3(1), dim3(1024), 0, s, part, npart, st, final, max_records);
    hipLaunchKernelGGL((k_line_scan<FASTQ, POS>), dim3((uint32_t)npart), dim3(PB), 0, s, line_len, line_head, st, part, line_dst,
                       line_start, max_records, offsets, name_pos, name_len);
    hipLaunchKernelGGL((k_parse_decide<FASTQ, POS>), dim3(1), dim3(1), 0, s, st, n, final, max_lines, nlpos, line_start, line_dst,
                       max_records, offsets, seq_capacity, res);
    hipLaunchKernelGGL((k_parse_copy<FASTQ, POS>), dim3(gc), dim3(PB), 0, s, st, max_records, seq_capacity, text, line_start,
                       line_len, line_head, line_dst, offsets, seq, qual);
    hipLaunchKernelGGL((k_parse_copy_long<FASTQ, POS>), dim3(1024), dim3(PB), 0, s, st, max_records, seq_capacity, text, n,
                       line_start, line_len, line_head, line_dst, offsets, seq, qual);
    return hipGetLastError();
}

hipError_t launch_parse(int fastq, const uint8_t* text, uint64_t n, int final, uint8_t* seq, uint8_t* qual,
                        uint64_t seq_capacity, uint64_t* offsets, uint64_t* name_pos, uint32_t* name_len,
                        uint64_t max_records, void* workspace, uint64_t max_lines, ParseResult* res, hipStream_t s)
{
    uint8_t* ws = static_cast<uint8_t*>(workspace);
    // a text of less than 4 GiB: every position in it and in the planes fits 32 bits, and the tables of positions
    // are half the bytes
    // (SKH_PARSE_POS64, read at every call: the 64-bit tables for a small text too -- tests of that path)
    if (n <= 0xffffffffull && getenv("SKH_PARSE_POS64") == nullptr)
        return fastq ? launch_parse_t<true, uint32_t>(text, n, final, seq, qual, seq_capacity, offsets, name_pos, name_len,
                                                      max_records, ws, max_lines, res, s)
                     : launch_parse_t<false, uint32_t>(text, n, final, seq, qual, seq_capacity, offsets, name_pos, name_len,
                                                       max_records, ws, max_lines, res, s);
    return fastq ? launch_parse_t<true, uint64_t>(text, n, final, seq, qual, seq_capacity, offsets, name_pos, name_len,
                                                  max_records, ws, max_lines, res, s)
                 : launch_parse_t<false, uint64_t>(text, n, final, seq, qual, seq_capacity, offsets, name_pos, name_len,
                                                   max_records, ws, max_lines, res, s);
}

// Workspace of the wrapped-FASTQ passes, behind the parser's own (all sections 256-byte aligned).
struct WrapLayout {
    uint64_t state, total, tmp, cP, cPlus, cNb, cRid, part, plus_idx, nb_idx, nxt0, nxtA, nxtB, mark, rec_start, rec_plus, rec_last, rec_S, end;
    WrapLayout(uint64_t base, uint64_t max_lines, uint64_t max_records)
    {
        const uint64_t Lc = max_lines + 4, Rc = max_records + 2;
        uint64_t o = align_up(base, 256);
        state = o; o = align_up(o + sizeof(WrapState), 256);
        total = o; o = align_up(o + 64, 256);
        tmp = o; o = align_up(o + Lc * 8, 256);
        cP = o; o = align_up(o + Lc * 8, 256);
        cPlus = o; o = align_up(o + Lc * 8, 256);
        cNb = o; o = align_up(o + Lc * 8, 256);
        cRid = o; o = align_up(o + Lc * 8, 256);
        part = o; o = align_up(o + scan64_partials(Lc > Rc ? Lc : Rc) * 8, 256);
        plus_idx = o; o = align_up(o + Lc * 4, 256);
        nb_idx = o; o = align_up(o + Lc * 4, 256);
        nxt0 = o; o = align_up(o + Lc * 4, 256);
        nxtA = o; o = align_up(o + Lc * 4, 256);
        nxtB = o; o = align_up(o + Lc * 4, 256);
        mark = o; o = align_up(o + Lc, 256);
        rec_start = o; o = align_up(o + Rc * 4, 256);
        rec_plus = o; o = align_up(o + Rc * 4, 256);
        rec_last = o; o = align_up(o + Rc * 4, 256);
        rec_S = o; o = align_up(o + Rc * 8, 256);
        end = o;
    }
};

uint64_t parse_wrapped_workspace_size(uint64_t text_bytes, uint64_t max_lines, uint64_t max_records)
{
    return WrapLayout(ParseLayout(text_bytes, max_lines).end, max_lines, max_records).end;
}

hipError_t launch_parse_wrapped(const uint8_t* text, uint64_t n, int final, uint8_t* seq, uint8_t* qual, uint64_t seq_capacity,
                                uint64_t* offsets, uint64_t* name_pos, uint32_t* name_len, uint64_t max_records, void* workspace,
                                uint64_t max_lines, ParseResult* res, hipStream_t s)
{
    uint8_t* ws = static_cast<uint8_t*>(workspace);
    const ParseLayout lay(n, max_lines);
    const WrapLayout wl(lay.end, max_lines, max_records);
    ParseState* st = reinterpret_cast<ParseState*>(ws + lay.state);
    uint32_t* tile_count = reinterpret_cast<uint32_t*>(ws + lay.tile_count);
    uint64_t* tile_base = reinterpret_cast<uint64_t*>(ws + lay.tile_base);
    uint64_t* tile_last = reinterpret_cast<uint64_t*>(ws + lay.tile_last);
    uint32_t* nl_tile = reinterpret_cast<uint32_t*>(ws + lay.nl_tile);
    uint64_t* nlpos = reinterpret_cast<uint64_t*>(ws + lay.nlpos);
    uint64_t* line_start = reinterpret_cast<uint64_t*>(ws + lay.line_start);
    uint32_t* line_len = reinterpret_cast<uint32_t*>(ws + lay.line_len);
    uint8_t* line_first = ws + lay.line_head;
    WrapState* w = reinterpret_cast<WrapState*>(ws + wl.state);
    uint64_t* total = reinterpret_cast<uint64_t*>(ws + wl.total);
    uint64_t* tmp = reinterpret_cast<uint64_t*>(ws + wl.tmp);
    uint64_t* cP = reinterpret_cast<uint64_t*>(ws + wl.cP);
    uint64_t* cPlus = reinterpret_cast<uint64_t*>(ws + wl.cPlus);
    uint64_t* cNb = reinterpret_cast<uint64_t*>(ws + wl.cNb);
    uint64_t* cRid = reinterpret_cast<uint64_t*>(ws + wl.cRid);
    uint64_t* part = reinterpret_cast<uint64_t*>(ws + wl.part);
    uint32_t* plus_idx = reinterpret_cast<uint32_t*>(ws + wl.plus_idx);
    uint32_t* nb_idx = reinterpret_cast<uint32_t*>(ws + wl.nb_idx);
    uint32_t* nxt0 = reinterpret_cast<uint32_t*>(ws + wl.nxt0);
    uint32_t* nxtA = reinterpret_cast<uint32_t*>(ws + wl.nxtA);
    uint32_t* nxtB = reinterpret_cast<uint32_t*>(ws + wl.nxtB);
    uint8_t* mark = ws + wl.mark;
    uint32_t* rec_start = reinterpret_cast<uint32_t*>(ws + wl.rec_start);
    uint32_t* rec_plus = reinterpret_cast<uint32_t*>(ws + wl.rec_plus);
    uint32_t* rec_last = reinterpret_cast<uint32_t*>(ws + wl.rec_last);
    uint64_t* rec_S = reinterpret_cast<uint64_t*>(ws + wl.rec_S);

    const uint64_t ntile = (n + TILE - 1) / TILE;
    const uint32_t gt = (uint32_t)(ntile ? ntile : 1);
    const uint64_t lines_cap = max_lines < n + 1 ? max_lines : n + 1;
    const uint64_t cap = lines_cap + 3;                    // the lines, and the three that stand for how a chain ends
    auto capped = [](uint64_t blocks) { return (uint32_t)(blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks)); };
    const uint32_t gl = capped((lines_cap + PB - 1) / PB), gw = capped((cap + WQ_B - 1) / WQ_B);
    hipError_t e = hipMemsetAsync(w, 0, sizeof(WrapState), s);
    if (e == hipSuccess) e = hipMemsetAsync(rec_S, 0, (max_records + 2) * 8, s);
    if (e != hipSuccess) return e;
    // the line table as for FASTA, with every line's first byte in place of the head flag
    hipLaunchKernelGGL(k_nl_index, dim3(gt), dim3(PB), 0, s, text, n, tile_count, tile_last, nl_tile);
    launch_tile_scan(tile_count, ntile, tile_base, reinterpret_cast<uint64_t*>(ws + lay.tile_part), st, n, final, text, max_lines, s);
    hipLaunchKernelGGL((k_line_table<false, uint64_t>), dim3(gl < 2048 ? gl : 2048), dim3(PB), 0, s, text, n, ntile, nl_tile, tile_count, tile_base,
                       tile_last, st, nlpos, line_start, line_len, line_first, 1, 0, nullptr, nullptr);
    uint64_t* sums[3] = {cP, cPlus, cNb};
    for (int kind = 0; kind < 3; kind++) {
        hipLaunchKernelGGL(k_wq_flags, dim3(gw), dim3(WQ_B), 0, s, st, line_len, line_first, kind, cap, tmp);
        if ((e = launch_scan64(tmp, cap, sums[kind], part, total, s)) != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_wq_index, dim3(gw), dim3(WQ_B), 0, s, st, line_len, line_first, cPlus, cNb, cap, plus_idx, nb_idx);
    hipLaunchKernelGGL(k_wq_next, dim3(gw), dim3(WQ_B), 0, s, st, final, line_len, line_first, cP, cPlus, cNb, plus_idx, nb_idx, cap,
                       nxt0, nxtA, mark);
    uint32_t rounds = 1;
    while ((1ull << rounds) < cap + 1) rounds++;
    uint32_t *a = nxtA, *b = nxtB;
    for (uint32_t r = 0; r < rounds; r++) {
        hipLaunchKernelGGL(k_wq_jump, dim3(gw), dim3(WQ_B), 0, s, st, cap, a, b, mark);
        uint32_t* t = a; a = b; b = t;
    }
    hipLaunchKernelGGL(k_wq_emit, dim3(gw), dim3(WQ_B), 0, s, st, n, final, nlpos, line_start, nxt0, mark, cap, tmp, w);
    if ((e = launch_scan64(tmp, cap, cRid, part, total, s)) != hipSuccess) return e;       // *total: records
    hipLaunchKernelGGL(k_wq_records, dim3(gw), dim3(WQ_B), 0, s, st, final, line_len, line_first, line_start, cP, cPlus, cNb, plus_idx,
                       nb_idx, cRid, tmp, cap, max_records, rec_start, rec_plus, rec_last, rec_S, name_pos, name_len, w);
    if ((e = launch_scan64(rec_S, max_records + 1, offsets, part, total + 1, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_wq_check, dim3(1), dim3(1), 0, s, st, max_lines, total, offsets, max_records, seq_capacity, w);
    hipLaunchKernelGGL(k_wq_copy, dim3(capped((lines_cap * 8 + WQ_B - 1) / WQ_B)), dim3(WQ_B), 0, s, st, w, text, line_start, line_len,
                       cP, cRid, rec_start, rec_plus, rec_last, offsets, cap, seq, qual);
    hipLaunchKernelGGL(k_wq_finish, dim3(1), dim3(1), 0, s, w, offsets, res);
    return hipGetLastError();
}

}  // namespace skh
