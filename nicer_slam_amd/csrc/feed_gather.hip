// feed_gather.hip -- the per-iteration batch of the resident frame stores in ONE launch:
//   out_f[i, k, :] = store_f[slots[i], sel[k], :]   for every field f (colour, monocular depth, normal, metric depth, mask)
//   uv[i, k]       = (sel[k] % width, sel[k] / width)
// Reference: SLAMDataset.__getitem__ + collate_fn (code/datasets/scene_dataset.py:214-275) index every image of every frame of the
// batch with sampling_idx on the host and upload the result; FrameFeed (nicer_slam_amd/feed.py) keeps the frames in HBM, where the same
// batch was 5 x b index_select launches (266 us per mapping iteration of 8 keyframes, profiles/r04_mapping_host.txt).
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/nicer_slam_amd.h"

namespace nsa {

constexpr int kMaxFeedFields = 8;
struct FeedArgs {
    const float* store[kMaxFeedFields];
    float* out[kMaxFeedFields];
    uint32_t ch[kMaxFeedFields];
    uint32_t n_fields;
    const int32_t* slots;
    const int64_t* sel;
    uint32_t b, n, width;
    uint64_t pixels;
    float* uv;
};

// one thread per (frame, sampled pixel); rows are 4-12 bytes at random pixels of a 26-MB frame: a latency-bound gather of b*n rows per
// field, far below any roofline at b*n = 8192 -- the point is one launch instead of 5 b
__global__ __launch_bounds__(256) void k_feed_gather(FeedArgs a) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (uint64_t)a.b * a.n) return;
    const uint32_t i = (uint32_t)(idx / a.n), k = (uint32_t)(idx - (uint64_t)i * a.n);
    const int64_t px = a.sel[k];
    const bool ok = px >= 0 && (uint64_t)px < a.pixels;              // an index outside the image poisons its rows (torch asserts)
    const uint64_t row = (uint64_t)a.slots[i] * a.pixels + (ok ? (uint64_t)px : 0);
    for (uint32_t f = 0; f < a.n_fields; ++f) {
        const uint32_t c = a.ch[f];
        const float* src = a.store[f] + row * c;
        float* dst = a.out[f] + idx * c;
        for (uint32_t j = 0; j < c; ++j) dst[j] = ok ? src[j] : __builtin_nanf("");
    }
    if (a.uv) {
        a.uv[idx * 2] = ok ? (float)(px % a.width) : __builtin_nanf("");
        a.uv[idx * 2 + 1] = ok ? (float)(px / a.width) : __builtin_nanf("");
    }
}

// up to 8 small float segments copied in ONE launch (the per-call inputs of a cached graph: pose, pixel batch, intrinsics): blockIdx.y =
// segment; a torch copy_ per tensor costs a launch and ~8 us of host time each
constexpr int kMaxCopySegs = 8;
struct CopyArgs {
    float* dst[kMaxCopySegs];
    const float* src[kMaxCopySegs];
    uint32_t n[kMaxCopySegs];
};
__global__ __launch_bounds__(256) void k_copy_segments(CopyArgs a) {
    const uint32_t s = blockIdx.y;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < a.n[s]; i += gridDim.x * 256) a.dst[s][i] = a.src[s][i];
}

}  // namespace nsa

extern "C" int nsa_copy_segments(const nsa_copy_seg_t* segs, uint32_t n_segs, nsa_stream_t stream) {
    using namespace nsa;
    if (!segs || n_segs == 0 || n_segs > kMaxCopySegs) return NSA_EBADARG;
    CopyArgs a{};
    uint32_t longest = 0;
    for (uint32_t i = 0; i < n_segs; ++i) {
        if (segs[i].n && (!segs[i].dst || !segs[i].src)) return NSA_EBADARG;
        a.dst[i] = segs[i].dst; a.src[i] = segs[i].src; a.n[i] = segs[i].n;
        if (segs[i].n > longest) longest = segs[i].n;
    }
    if (longest == 0) return NSA_OK;
    uint32_t bx = (longest + 255) / 256;
    if (bx > 64) bx = 64;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_copy_segments, dim3(bx, n_segs), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? NSA_OK : NSA_ELAUNCH;
}

extern "C" int nsa_feed_gather(const nsa_feed_field_t* fields, uint32_t n_fields, const int32_t* slots, uint32_t b, const int64_t* sel,
                               uint32_t n, uint64_t pixels, uint32_t width, float* uv, nsa_stream_t stream) {
    using namespace nsa;
    if (!fields || n_fields == 0 || n_fields > kMaxFeedFields || !slots || !sel || pixels == 0 || width == 0) return NSA_EBADARG;
    if (b == 0 || n == 0) return NSA_OK;
    FeedArgs a{};
    for (uint32_t f = 0; f < n_fields; ++f) {
        if (!fields[f].store || !fields[f].out || fields[f].channels == 0 || fields[f].channels > 16) return NSA_EBADARG;
        a.store[f] = fields[f].store; a.out[f] = fields[f].out; a.ch[f] = fields[f].channels;
    }
    a.n_fields = n_fields; a.slots = slots; a.sel = sel; a.b = b; a.n = n; a.width = width; a.pixels = pixels; a.uv = uv;
    (void)hipGetLastError();
    const uint64_t items = (uint64_t)b * n;
    hipLaunchKernelGGL(k_feed_gather, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? NSA_OK : NSA_ELAUNCH;
}
