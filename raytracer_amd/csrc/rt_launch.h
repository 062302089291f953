// Launchers of the pipeline kernels (defined in rt_kernels.hip), called by the C-ABI layer.
#pragma once
#include "rt_kernels.h"

namespace rt {

struct LevelSet { Level lv[RT_MAX_LEVELS]; };

// Rays of the pruning traversal whose answer is suspect (see traverse in rt_device.h), per kernel kind: a count and a
// ticket word inside the counter block of the call and a list of `cap` entries (node numbers for the nearest-hit
// kernel; item << 4 | ray for the light-ray kernel).  The second launch empties the queue and zeroes count and ticket
// itself, so a frame that queues nothing costs two tiny launches and no memset.  A queue that overflows sets the scene's
// fault word (reported by the next call that checks it).
struct RedoQueue {
    uint32_t* count_extend; uint32_t* ticket_extend; uint32_t* total_extend; uint32_t* list_extend;
    uint32_t* count_shadow; uint32_t* ticket_shadow; uint32_t* total_shadow; unsigned long long* list_shadow;
    uint32_t cap;
};

// counters[RT_KIDS_WORD0 + q * 32] (q = 0..63, summed by the caller) += number of children the level will spawn
// work (may be NULL): work[0] += inner nodes fetched, work[1] += triangles tested
// redo (pruning traversal only, may be NULL): nodes whose hit is a triangle the reference never looks at are queued there
// and walked again, every candidate vetted, by a second small launch that redoes their results (RedoQueue)
void launch_extend(bool exact, const DScene& sc, const RenderParams& P, const Level& L, int64_t n, uint32_t* counters,
                   unsigned long long* work, const struct RedoQueue* redo, hipStream_t s);
// traced (may be NULL): light rays traced, accumulated in RT_TRACED_SLOTS counters RT_TRACED_STRIDE apart (one
// address would serialise the per-wave atomics in one L2 channel); the caller sums the slots.  The third 64-bit word
// of each slot counts the light rays left out under rt_options.skip_unlit
enum { RT_TRACED_SLOTS = 64, RT_TRACED_STRIDE = 16,
       RT_KIDS_WORD0 = 16 + 2 };     // the children counts share the slots' 128-byte lines: u32 word 2 of each
void launch_shadow(bool exact, const DScene& sc, const RenderParams& P, const Level& L, int64_t n_items, unsigned long long* traced,
                   unsigned long long* work, const struct RedoQueue* redo, hipStream_t s);
void launch_spawn(const DScene& sc, const RenderParams& P, const Level& L, const Level& N, int64_t n, uint32_t* cursor, hipStream_t s);
void launch_replay(const DScene& sc, const RenderParams& P, const LevelSet& ls, int64_t n_pixels, float* rgb_out, hipStream_t s);
// replay of a frame with a single level, one lane per sample (spp^2 <= 64); n_slots = pixel slots incl. tile padding
void launch_replay_flat(const DScene& sc, const RenderParams& P, const LevelSet& ls, int64_t n_slots, float* rgb_out, hipStream_t s);
// the replay in two passes: one lane per level-0 node shades its sample's tree into draw records, then one lane per
// pixel adds them in pop order (levels must carry draw arrays, level 0 also `first`)
void launch_shade_sum(const DScene& sc, const RenderParams& P, const LevelSet& ls, int64_t n_nodes0, int64_t n_pixels, float* rgb_out, hipStream_t s);
// vet: a suspect answer of the pruning traversal is walked again at once, every candidate vetted (see traverse)
void launch_trace(bool any, bool exact, bool vet, const DScene& sc, const float* org, const float* dir, float t0, float t1, int64_t n,
                  float* t_out, int32_t* tri_out, float* nrm_out, uint32_t* cn, uint32_t* ct, hipStream_t s);
// the pruning traversal's nodes get their child references packed into their box planes (rt_device.h packed_ref)
void launch_pack_refs(Node64* nodes, int64_t n, hipStream_t s);
// the pruning traversal's triangle records get, in their first spare word, the number of the reference's leaf that holds
// the triangle (leaf_of_id: by caller's triangle id; rt_device.h ref_leaf_box_hit)
void launch_set_ref_leaf(Tri48* tris, const TriInfo* info, const int32_t* leaf_of_id, int64_t n, hipStream_t s);
void launch_resolve_u8(const float* rgb, int W, int H, uint8_t* out, hipStream_t s);

}  // namespace rt
