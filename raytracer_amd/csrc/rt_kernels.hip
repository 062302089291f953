// __global__ wrappers and launchers of the wavefront pipeline (bodies: rt_kernels.h).
#include <algorithm>

#include "rt_launch.h"

namespace rt {
namespace {

constexpr int BLOCK = RT_BLOCK;

// LDS part of the traversal stacks of one workgroup: [RT_LDS_SLOTS][BLOCK]
#define RT_DECLARE_STACK(st)                                                          \
    __shared__ int32_t lds_stack_[RT_LDS_SLOTS * BLOCK];                              \
    int32_t stack_spill_[RT_STACK - RT_LDS_SLOTS];                                    \
    StackLds st;                                                                      \
    st.lds = (RT_LDS_PTR int32_t*)lds_stack_ + threadIdx.x; st.spill = stack_spill_; st.fault = sc.fault; st.sp = 0
// the same for a tree no deeper than RT_LDS_SLOTS (DEEP == false: no private part)
#define RT_DECLARE_STACK_T(st, DEEP)                                                  \
    __shared__ int32_t lds_stack_[RT_LDS_SLOTS * BLOCK];                              \
    int32_t stack_spill_[DEEP ? RT_STACK - RT_LDS_SLOTS : 1];                         \
    StackT<DEEP> st;                                                                  \
    st.lds = (RT_LDS_PTR int32_t*)lds_stack_ + threadIdx.x; st.spill = stack_spill_; st.fault = sc.fault; st.sp = 0

// Sum over the 64 lanes, the same value returned to every lane.  Lane-to-lane adds inside the vector ALU (a running sum
// along each row of 16 lanes, then the row totals carried into the following rows) instead of six exchanges through the
// LDS crossbar; the total ends up in lane 63.
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}

// Workgroups are handed to the eight XCDs (each with its own L2) in turn: hardware workgroup b runs on XCD b % 8.  With
// one wave per workgroup, eight consecutive waves - neighbours in the image, which read neighbouring queue records and
// the same nodes - would land in eight different L2s.  Inside every run of 64 workgroups the kernels therefore swap the
// two 3-bit digits of b: XCD x takes the 8 CONSECUTIVE work units 8x .. 8x+7 of the run.  (Giving each XCD one
// contiguous eighth of the whole launch is no good: the parts of an image differ too much in cost - 1.03 -> 1.95 ms;
// 32 consecutive units per XCD instead of 8 made no difference.)
// Grids are rounded up to a multiple of 64 (xcd_grid); every kernel already ignores units beyond its data.
__device__ __forceinline__ uint32_t xcd_block() {
    const uint32_t b = blockIdx.x;
    return (b & ~63u) | ((b & 7u) << 3) | ((b >> 3) & 7u);
}

// work[0] += inner nodes fetched, work[1] += triangles tested (COUNT builds only)
template <bool COUNT>
__device__ __forceinline__ void add_work(unsigned long long* work, uint32_t nn, uint32_t nt) {
    if (COUNT) {
        nn = wave_sum(nn); nt = wave_sum(nt);
        if ((threadIdx.x & 63) == 0) { atomicAdd(&work[0], (unsigned long long)nn); atomicAdd(&work[1], (unsigned long long)nt); }
    }
}

// appends one entry to a redo list (one atomic per suspect ray: about one ray in a hundred thousand)
template <class T>
__device__ __forceinline__ void redo_push(uint32_t* count, T* list, uint32_t cap, T entry, uint32_t* fault) {
    const uint32_t at = atomicAdd(count, 1u);
    if (at < cap) list[at] = entry; else *fault = 2u;
}

// The nearest-hit kernels of the pruning traversal need 65 - 67 vector registers when left to themselves, one or three too
// many for eight waves per SIMD; told to fit, the compiler fits them into 64 without scratch, and the eighth wave hides
// that much more latency: mirror room nearest-hit kernel 23.2 -> 20.8 ms per frame, 10 M field 267 -> 231.  (The
// reference-arithmetic instantiations need 94 and stay as they are.)
#define RT_EXTEND_WAVES __attribute__((amdgpu_waves_per_eu(EXACT ? 1 : 8, 8)))
// CHECK (pruning traversal only): the winner is checked against its reference leaf's box and a suspect ray queued
template <bool EXACT, bool COUNT, bool DEEP = true, bool CHECK = false>
__global__ __launch_bounds__(BLOCK) RT_EXTEND_WAVES void k_extend(DScene sc, RenderParams P, Level L, int64_t n, uint32_t* counters, unsigned long long* work, RedoQueue redo) {
    const int64_t i = (int64_t)xcd_block() * BLOCK + threadIdx.x;
    uint32_t kids = 0, nn = 0, nt = 0;
    RT_DECLARE_STACK_T(st, DEEP);
    bool suspect = false;
    if (i < n) kids = (uint32_t)body_extend<EXACT>(sc, P, L, st, i, COUNT ? &nn : nullptr, COUNT ? &nt : nullptr, (!EXACT && CHECK) ? &suspect : nullptr);
    if (!EXACT && CHECK && suspect) redo_push(redo.count_extend, redo.list_extend, redo.cap, (uint32_t)i, sc.fault);
    if (!P.can_spawn) { add_work<COUNT>(work, nn, nt); return; }     // no material of the scene spawns rays: nothing to count
    kids = wave_sum(kids);
    // one atomic per wave on ONE word would queue up in a single L2 channel (about 8 ns each): 64 words 128 bytes apart
    if ((threadIdx.x & 63) == 0 && kids) atomicAdd(&counters[RT_KIDS_WORD0 + (blockIdx.x & (RT_TRACED_SLOTS - 1)) * RT_TRACED_STRIDE * 2], kids);
    add_work<COUNT>(work, nn, nt);
}

// One lane per light ray: the spr+1 rays of an item sit in consecutive lanes (a wave takes 64 / (spr+1) items and
// leaves the remaining lanes idle), so the lanes of a wave trace rays of few neighbouring origins and the wave is done
// when its slowest ray is - a lane that looped over the item's rays would wait for the slowest lane once per ray.
// The item's count is the population count of the wave's "reached the light" mask over the item's lanes.
// Second walk of the queued nodes, every candidate vetted: all of a node's results are written again, and the level's
// count of children is corrected by the difference (as a wrapped 32-bit sum: the host adds the slots as signed numbers).
// A handful of workgroups stride over the list; the last one to finish zeroes count and ticket for the next use.
constexpr int REDO_BLOCKS = 32;
template <class F>
__device__ __forceinline__ void redo_drain(uint32_t* count, uint32_t* ticket, uint32_t* total, uint32_t cap, F&& each) {
    const uint32_t n = min(*reinterpret_cast<volatile uint32_t*>(count), cap);
    if (n && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(total, n);      // for the call's statistics
    for (uint32_t k = blockIdx.x * BLOCK + threadIdx.x; k < n; k += gridDim.x * BLOCK) each(k);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(ticket, 1u) == gridDim.x - 1) { *count = 0u; *ticket = 0u; }
}
__global__ __launch_bounds__(BLOCK) void k_extend_redo(DScene sc, RenderParams P, Level L, uint32_t* counters, RedoQueue redo) {
    RT_DECLARE_STACK(st);
    redo_drain(redo.count_extend, redo.ticket_extend, redo.total_extend, redo.cap, [&](uint32_t k) {
        const int64_t i = redo.list_extend[k];
        const int before = (P.can_spawn && res_tri(L, i) >= 0) ? res_kid_count(L, i) : 0;
        const int after = body_extend<false, StackLds, true>(sc, P, L, st, i);
        if (P.can_spawn && after != before) atomicAdd(&counters[RT_KIDS_WORD0], (uint32_t)(after - before));
    });
}
__global__ __launch_bounds__(BLOCK) void k_shadow_redo(DScene sc, RenderParams P, Level L, RedoQueue redo) {
    RT_DECLARE_STACK(st);
    redo_drain(redo.count_shadow, redo.ticket_shadow, redo.total_shadow, redo.cap, [&](uint32_t k) {
        const unsigned long long e = redo.list_shadow[k];
        const int64_t w = (int64_t)(e >> 4);
        ShadowItem it;
        it.idle = false;
        // the first walk found the ray occluded (by the suspect triangle): if the vetted walk reaches the light, the item's
        // count is one short
        if (shadow_item(sc, P, L, w, it) && shadow_ray<false, StackLds, true>(sc, P, it, st, (int)(e & 15))) bump_unocc(sc, P, L, w);
    });
}

template <bool EXACT, bool COUNT, bool DEEP = true, bool CHECK = false>
__global__ __launch_bounds__(BLOCK) void k_shadow(DScene sc, RenderParams P, Level L, int64_t n_items, FastDiv div_nodes, uint32_t n_nodes, unsigned long long* traced, unsigned long long* work, RedoQueue redo) {
    const int R = P.spr + 1, per_wave = 64 / R;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)xcd_block() * BLOCK + threadIdx.x) >> 6;
    const int sub = (int)fdiv((uint32_t)lane, P.div_rays), k = lane - sub * R;
    // items are taken light by light (all nodes toward light 0, then light 1, ...): the rays of a wave then go toward
    // the same light and share a direction octant; item number w = node * n_lights + light is what the records use
    const int64_t v = wave * per_wave + sub;
    const uint32_t light = fdiv((uint32_t)v, div_nodes);
    const int64_t w = (int64_t)((uint32_t)v - light * n_nodes) * sc.n_lights + light;
    uint32_t did = 0, nn = 0, nt = 0;
    bool reached = false;
    RT_DECLARE_STACK_T(st, DEEP);
    ShadowItem it;
    it.idle = false;
    const bool item = sub < per_wave && v < n_items;
    if (item && shadow_item(sc, P, L, w, it)) {
        did = 1;
        bool suspect = false;
        reached = shadow_ray<EXACT>(sc, P, it, st, k, COUNT ? &nn : nullptr, COUNT ? &nt : nullptr, (!EXACT && CHECK) ? &suspect : nullptr);
        if (!EXACT && CHECK && suspect) redo_push(redo.count_shadow, redo.list_shadow, redo.cap, ((unsigned long long)w << 4) | (unsigned long long)k, sc.fault);
    }
    const unsigned long long mask = __ballot(reached);
    if (did && k == 0) store_unocc(sc, P, L, w, (uint32_t)__popcll((mask >> lane) & ((1ull << R) - 1ull)));
    // below level 0 the replay learns from the count byte itself that the companion ray missed (RT_NO_COMPANION_HIT)
    else if (item && k == 0 && !L.implicit) store_unocc(sc, P, L, w, RT_NO_COMPANION_HIT);
    if (traced) {
        did = (uint32_t)__popcll(__ballot(did != 0));
        if (lane == 0 && did) atomicAdd(&traced[(blockIdx.x & (RT_TRACED_SLOTS - 1)) * RT_TRACED_STRIDE], (unsigned long long)did);
        // light rays the reference would have traced here and this kernel does not (third 64-bit word of the slot)
        const uint32_t idle = (uint32_t)__popcll(__ballot(it.idle));
        if (lane == 0 && idle) atomicAdd(&traced[(blockIdx.x & (RT_TRACED_SLOTS - 1)) * RT_TRACED_STRIDE + 2], (unsigned long long)idle);
    }
    add_work<COUNT>(work, nn, nt);
}

// The children of a workgroup's nodes get consecutive places, in node order, from ONE atomic per workgroup of 1024
// nodes (an atomic per wave on the one cursor word queues up in a single L2 channel, about 8 ns each - more than the
// rest of the kernel): neighbouring nodes of the next level are then children of neighbouring pixels (rays that stay
// together in the tree).  The records are written one CHILD per lane: every parent lane notes (lane, child number) for
// each of its places in a small LDS table, and the wave walks its places 64 at a time, so that consecutive lanes write
// consecutive 16-byte records.
constexpr int SPAWN_BLOCK = 1024;
__global__ __launch_bounds__(SPAWN_BLOCK) void k_spawn(DScene sc, RenderParams P, Level L, Level N, int64_t n, uint32_t* cursor) {
    __shared__ uint16_t owner_[SPAWN_BLOCK / 64][64 * 16];       // up to spr + 1 <= 16 children per node
    __shared__ uint32_t wave_total_[SPAWN_BLOCK / 64], wave_base_[SPAWN_BLOCK / 64];
    const int64_t i = (int64_t)blockIdx.x * SPAWN_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t cnt = (i < n && res_tri(L, i) >= 0) ? (uint32_t)res_kid_count(L, i) : 0u;
    uint32_t incl = cnt;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    const uint32_t total = __shfl(incl, 63, 64);
    if (lane == 0) wave_total_[wave] = total;
    const uint32_t excl = incl - cnt;
    for (uint32_t k = 0; k < cnt; ++k) owner_[wave][excl + k] = (uint16_t)(lane | (k << 8));
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t sum = 0;
        for (int w = 0; w < SPAWN_BLOCK / 64; ++w) { wave_base_[w] = sum; sum += wave_total_[w]; }
        const uint32_t b0 = sum ? atomicAdd(cursor, sum) : 0u;
        for (int w = 0; w < SPAWN_BLOCK / 64; ++w) wave_base_[w] += b0;
    }
    __syncthreads();
    const uint32_t base = wave_base_[wave];
    if (cnt) set_kids(L, i, (int32_t)(base + excl), (int)cnt);
    const int64_t i0 = i - lane;                           // the wave's first node
    for (uint32_t c = (uint32_t)lane; c < total; c += 64) {
        const uint32_t o = owner_[wave][c];
        const int pl = (int)(o & 0xffu), k = (int)(o >> 8);
        spawn_child(sc, P, L, N, base + c - (uint32_t)k, base + c, i0 + pl, k);
    }
}

// LDS and scratch parts of the replay's frames (one wave per workgroup)
#define RT_DECLARE_FRAMES(fr)                                                          \
    static_assert(BLOCK == 64, "the frame store is laid out for one wave per workgroup"); \
    __shared__ rt_f4 lds_frames_[RT_FRAME_LDS_LEVELS * 2 * 64];                        \
    RFrame frame_spill_[RT_MAX_LEVELS];                                                \
    FrameStore fr;                                                                     \
    fr.lds = (RT_LDS_PTR rt_f4*)lds_frames_ + threadIdx.x; fr.spill = frame_spill_

__global__ __launch_bounds__(BLOCK) void k_replay(DScene sc, RenderParams P, LevelSet ls, int64_t n_pixels, float* rgb_out) {
    const int64_t pl = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    RT_DECLARE_FRAMES(fr);
    if (pl < n_pixels) body_replay(sc, P, ls.lv, pl, rgb_out, fr);
}

// Replay of a frame that has only level 0 (no material spawns rays), one lane per SAMPLE: the samples of a pixel sit in
// consecutive lanes (64 / spp^2 pixels per wave), every load is coalesced in node order, and the pixel's lanes then add
// their colours in the reference's pop order (last sample first) through lane shuffles - the same additions in the
// same order as the lane-per-pixel loop, with spp^2 times the lanes.
__global__ __launch_bounds__(BLOCK) void k_replay_flat(DScene sc, RenderParams P, LevelSet ls, int64_t n_slots, float* rgb_out) {
    const int spp2 = P.spp * P.spp, per_wave = 64 / spp2;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)xcd_block() * BLOCK + threadIdx.x) >> 6;
    const int sub = (int)fdiv((uint32_t)lane, P.div_spp2), smp = lane - sub * spp2;
    const int64_t slot = wave * per_wave + sub;
    const bool active = sub < per_wave && slot < n_slots;
    V3 c = v3(0, 0, 0);
    const int drew = (active && replay_flat_sample(sc, P, ls.lv[0], slot * spp2 + smp, c)) ? 1 : 0;
    V3 sum = v3(0, 0, 0), wsum = v3(0, 0, 0);
    for (int k = spp2 - 1; k >= 0; --k) {
        const int src = lane - smp + k;
        const float cx = __shfl(c.x, src, 64), cy = __shfl(c.y, src, 64), cz = __shfl(c.z, src, 64);
        if (__shfl(drew, src, 64)) { sum = vadd(sum, v3(cx, cy, cz)); wsum = vadd(wsum, v3(1.f, 1.f, 1.f)); }   // a primary ray weighs 1
    }
    if (!active || smp != 0) return;
    const uint32_t tile = (uint32_t)(slot >> 4), tile_y = fdiv(tile, P.div_tiles_x), tile_x = tile - tile_y * (uint32_t)((P.rw + 3) >> 2);
    const int q = (int)(slot & 15);
    const int px = (int)tile_x * 4 + ((q & 1) | ((q >> 1) & 2)), py = (int)tile_y * 4 + (((q >> 1) & 1) | ((q >> 2) & 2));
    if (px >= P.rw || py >= P.band_rows) return;
    const int64_t pl = (int64_t)py * P.rw + px;
    rgb_out[pl * 3 + 0] = sum.x / wsum.x;
    rgb_out[pl * 3 + 1] = sum.y / wsum.y;
    rgb_out[pl * 3 + 2] = sum.z / wsum.z;
}

__global__ __launch_bounds__(BLOCK) void k_shade(DScene sc, RenderParams P, LevelSet ls, int64_t n) {
    const int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    RT_DECLARE_FRAMES(fr);
    if (g < n) body_shade(sc, P, ls.lv, g, fr);
}

__global__ __launch_bounds__(BLOCK) void k_sum(RenderParams P, LevelSet ls, int64_t n_pixels, float* rgb_out) {
    const int64_t pl = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (pl < n_pixels) body_sum(P, ls.lv, pl, rgb_out);
}

template <bool ANY, bool EXACT>
__global__ __launch_bounds__(BLOCK) void k_trace(DScene sc, const float* org, const float* dir, float t0, float t1, int64_t n,
                                                 float* t_out, int32_t* tri_out, float* nrm_out, uint32_t* cn, uint32_t* ct, int vet) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    RT_DECLARE_STACK(st);
    if (i < n) body_trace<ANY, EXACT>(sc, org, dir, t0, t1, i, t_out, tri_out, nrm_out, cn, ct, st, vet != 0);
}

// kernels without a traversal stack or a tail of slow waves keep four waves per workgroup (fewer, fuller workgroups)
constexpr int FLAT_BLOCK = 256;
__global__ __launch_bounds__(FLAT_BLOCK) void k_resolve_u8(const float* rgb, int W, int H, int64_t n, uint8_t* out) {
    const int64_t i = (int64_t)blockIdx.x * FLAT_BLOCK + threadIdx.x;
    if (i < n) body_resolve_u8(rgb, W, H, i, out);
}

// Writes byte k of a child reference into the low mantissa byte of plane k of that child's box (see packed_ref), moving
// a lower plane down and an upper plane up to the nearest float with that low byte: at most 511 units in the last place,
// never inwards.  Works on the order-preserving integer image of a float (all bits flipped for negative numbers, the
// sign bit set for the others), in which a block of 256 consecutive values shares its upper 24 bits and its sign.
__device__ __forceinline__ uint32_t float_key(float f) { const uint32_t b = (uint32_t)__float_as_int(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float key_float(uint32_t k) { return __int_as_float((int)((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k)); }
__device__ __forceinline__ float plane_with_byte(float f, uint32_t byte, bool upwards) {
    if (!(fabsf(f) < 3.0e38f)) f = (f > 0.f) ? 3.0e38f : -3.0e38f;        // planes of an absent child's box
    const uint32_t k = float_key(f);
    auto in_block = [&](uint32_t blk) { return (blk << 8) | ((blk & 0x800000u) ? byte : (~byte & 0xffu)); };
    uint32_t blk = k >> 8, c = in_block(blk);
    if (upwards ? (c < k) : (c > k)) c = in_block(upwards ? blk + 1u : blk - 1u);
    return key_float(c);
}
__global__ __launch_bounds__(FLAT_BLOCK) void k_pack_refs(Node64* nodes, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * FLAT_BLOCK + threadIdx.x;
    if (i >= n) return;
    Node64 nd = nodes[i];
    const uint32_t refs[2] = {(uint32_t)nd.lref_fine, (uint32_t)nd.rref_fine};
    for (int s = 0; s < 2; ++s) {
        float* b = s ? nd.rbox : nd.lbox;
        for (int k = 0; k < 4; ++k) b[k] = plane_with_byte(b[k], (refs[s] >> (8 * k)) & 0xffu, /*upwards*/ k == 3);
    }
    nodes[i] = nd;
}

__global__ __launch_bounds__(FLAT_BLOCK) void k_set_ref_leaf(Tri48* tris, const TriInfo* info, const int32_t* leaf_of_id, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * FLAT_BLOCK + threadIdx.x;
    if (i < n) tris[i].run = leaf_of_id[info[i].id];
}

inline unsigned blocks(int64_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }
inline unsigned xcd_grid(int64_t n) { return (blocks(n) + 63u) & ~63u; }      // for the kernels that use xcd_block()

}  // namespace

void launch_extend(bool exact, const DScene& sc, const RenderParams& P, const Level& L, int64_t n, uint32_t* counters,
                   unsigned long long* work, const RedoQueue* redo, hipStream_t s) {
    if (n <= 0) return;
    const unsigned g = xcd_grid(n);
    RedoQueue q{};
    if (redo && !exact) q = *redo;
    if (work) {
        if (exact) k_extend<true, true><<<g, BLOCK, 0, s>>>(sc, P, L, n, counters, work, q);
        else if (q.cap) k_extend<false, true, true, true><<<g, BLOCK, 0, s>>>(sc, P, L, n, counters, work, q);
        else k_extend<false, true><<<g, BLOCK, 0, s>>>(sc, P, L, n, counters, work, q);
    } else {
        if (exact) k_extend<true, false><<<g, BLOCK, 0, s>>>(sc, P, L, n, counters, nullptr, q);
        else if (sc.shallow && q.cap) k_extend<false, false, false, true><<<g, BLOCK, 0, s>>>(sc, P, L, n, counters, nullptr, q);
        else if (sc.shallow) k_extend<false, false, false><<<g, BLOCK, 0, s>>>(sc, P, L, n, counters, nullptr, q);
        else if (q.cap) k_extend<false, false, true, true><<<g, BLOCK, 0, s>>>(sc, P, L, n, counters, nullptr, q);
        else k_extend<false, false><<<g, BLOCK, 0, s>>>(sc, P, L, n, counters, nullptr, q);
    }
    if (q.cap) k_extend_redo<<<REDO_BLOCKS, BLOCK, 0, s>>>(sc, P, L, counters, q);
}
void launch_shadow(bool exact, const DScene& sc, const RenderParams& P, const Level& L, int64_t n_items, unsigned long long* traced,
                   unsigned long long* work, const RedoQueue* redo, hipStream_t s) {
    if (n_items <= 0) return;
    const int64_t per_wave = 64 / (P.spr + 1);
    const unsigned g = xcd_grid((n_items + per_wave - 1) / per_wave * 64);
    const uint32_t n_nodes = (uint32_t)(n_items / std::max(sc.n_lights, 1));
    const FastDiv dn = make_fastdiv(n_nodes);
    RedoQueue q{};
    if (redo && !exact) q = *redo;
    if (work) {
        if (exact) k_shadow<true, true><<<g, BLOCK, 0, s>>>(sc, P, L, n_items, dn, n_nodes, traced, work, q);
        else if (q.cap) k_shadow<false, true, true, true><<<g, BLOCK, 0, s>>>(sc, P, L, n_items, dn, n_nodes, traced, work, q);
        else k_shadow<false, true><<<g, BLOCK, 0, s>>>(sc, P, L, n_items, dn, n_nodes, traced, work, q);
    } else {
        if (exact) k_shadow<true, false><<<g, BLOCK, 0, s>>>(sc, P, L, n_items, dn, n_nodes, traced, nullptr, q);
        else if (sc.shallow && q.cap) k_shadow<false, false, false, true><<<g, BLOCK, 0, s>>>(sc, P, L, n_items, dn, n_nodes, traced, nullptr, q);
        else if (sc.shallow) k_shadow<false, false, false><<<g, BLOCK, 0, s>>>(sc, P, L, n_items, dn, n_nodes, traced, nullptr, q);
        else if (q.cap) k_shadow<false, false, true, true><<<g, BLOCK, 0, s>>>(sc, P, L, n_items, dn, n_nodes, traced, nullptr, q);
        else k_shadow<false, false><<<g, BLOCK, 0, s>>>(sc, P, L, n_items, dn, n_nodes, traced, nullptr, q);
    }
    if (q.cap) k_shadow_redo<<<REDO_BLOCKS, BLOCK, 0, s>>>(sc, P, L, q);
}
void launch_spawn(const DScene& sc, const RenderParams& P, const Level& L, const Level& N, int64_t n, uint32_t* cursor, hipStream_t s) {
    if (n > 0) k_spawn<<<(unsigned)((n + SPAWN_BLOCK - 1) / SPAWN_BLOCK), SPAWN_BLOCK, 0, s>>>(sc, P, L, N, n, cursor);
}
void launch_replay(const DScene& sc, const RenderParams& P, const LevelSet& ls, int64_t n_pixels, float* rgb_out, hipStream_t s) {
    if (n_pixels > 0) k_replay<<<blocks(n_pixels), BLOCK, 0, s>>>(sc, P, ls, n_pixels, rgb_out);
}
void launch_replay_flat(const DScene& sc, const RenderParams& P, const LevelSet& ls, int64_t n_slots, float* rgb_out, hipStream_t s) {
    if (n_slots <= 0) return;
    const int64_t per_wave = 64 / (P.spp * P.spp);
    k_replay_flat<<<xcd_grid((n_slots + per_wave - 1) / per_wave * 64), BLOCK, 0, s>>>(sc, P, ls, n_slots, rgb_out);
}
void launch_shade_sum(const DScene& sc, const RenderParams& P, const LevelSet& ls, int64_t n_nodes0, int64_t n_pixels, float* rgb_out, hipStream_t s) {
    if (n_nodes0 > 0) k_shade<<<blocks(n_nodes0), BLOCK, 0, s>>>(sc, P, ls, n_nodes0);
    if (n_pixels > 0) k_sum<<<blocks(n_pixels), BLOCK, 0, s>>>(P, ls, n_pixels, rgb_out);
}
void launch_trace(bool any, bool exact, bool vet, const DScene& sc, const float* org, const float* dir, float t0, float t1, int64_t n,
                  float* t_out, int32_t* tri_out, float* nrm_out, uint32_t* cn, uint32_t* ct, hipStream_t s) {
    if (n <= 0) return;
    const unsigned g = blocks(n);
    if (any && exact) k_trace<true, true><<<g, BLOCK, 0, s>>>(sc, org, dir, t0, t1, n, t_out, tri_out, nrm_out, cn, ct, vet ? 1 : 0);
    else if (any) k_trace<true, false><<<g, BLOCK, 0, s>>>(sc, org, dir, t0, t1, n, t_out, tri_out, nrm_out, cn, ct, vet ? 1 : 0);
    else if (exact) k_trace<false, true><<<g, BLOCK, 0, s>>>(sc, org, dir, t0, t1, n, t_out, tri_out, nrm_out, cn, ct, vet ? 1 : 0);
    else k_trace<false, false><<<g, BLOCK, 0, s>>>(sc, org, dir, t0, t1, n, t_out, tri_out, nrm_out, cn, ct, vet ? 1 : 0);
}
void launch_pack_refs(Node64* nodes, int64_t n, hipStream_t s) {
    if (n > 0) k_pack_refs<<<(unsigned)((n + FLAT_BLOCK - 1) / FLAT_BLOCK), FLAT_BLOCK, 0, s>>>(nodes, n);
}
void launch_set_ref_leaf(Tri48* tris, const TriInfo* info, const int32_t* leaf_of_id, int64_t n, hipStream_t s) {
    if (n > 0) k_set_ref_leaf<<<(unsigned)((n + FLAT_BLOCK - 1) / FLAT_BLOCK), FLAT_BLOCK, 0, s>>>(tris, info, leaf_of_id, n);
}
void launch_resolve_u8(const float* rgb, int W, int H, uint8_t* out, hipStream_t s) {
    const int64_t n = (int64_t)W * H * 3;
    if (n > 0) k_resolve_u8<<<(unsigned)((n + FLAT_BLOCK - 1) / FLAT_BLOCK), FLAT_BLOCK, 0, s>>>(rgb, W, H, n, out);
}

}  // namespace rt
