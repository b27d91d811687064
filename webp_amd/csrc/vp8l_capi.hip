// vp8l_capi.hip -- host-memory entry points of the C ABI (include/vp8l_amd.h) and shared host
// helpers.  The host-memory forms stage images through HBM in chunks, overlapping uploads, kernels and
// downloads on separate HIP streams, and call the device-memory batch forms; all codec work happens in
// the kernels of vp8l_encode.hip / vp8l_decode.hip.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "vp8l_amd.h"
#include "vp8l_common.h"
#include "vp8l_host.h"
#include "vp8l_split.h"

namespace vp8l {

std::string &last_error_slot() {
  static thread_local std::string s;
  return s;
}

bool hip_ok(hipError_t e, const char *what) {
  if (e == hipSuccess) return true;
  last_error_slot() = std::string(what) + ": " + hipGetErrorString(e);
  (void)hipGetLastError();
  return false;
}

int select_device(int device, int *chosen) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    last_error_slot() = "no HIP device available";
    (void)hipGetLastError();
    return VP8L_ERR_NO_DEVICE;
  }
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) return VP8L_ERR_NO_DEVICE;
  }
  if (device >= count) return VP8L_ERR_BAD_ARGUMENT;
  if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return VP8L_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (!hip_ok(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) return VP8L_ERR_NO_DEVICE;
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
    last_error_slot() = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
    return VP8L_ERR_NO_DEVICE;
  }
  if (chosen) *chosen = device;
  return VP8L_OK;
}

namespace {

// The host-memory entry points cut a batch into chunks and run them through a few lanes per device.  A lane
// owns an encoder or decoder context (with its HIP stream and workspace), staging buffers in HBM and a copy
// stream; a host thread per lane uploads a chunk, runs the kernels and downloads the results, so the upload of
// one chunk, the kernels of another and the download of a third overlap (copy engines and CUs work side by
// side).  Staging buffers are kept between calls (grow-only): hipMalloc / hipFree synchronise the device.
// Measured on a 128 x 3840x2160 batch from page-locked memory: 3 lanes x 128 MiB move 22 GB/s, 8 x 64 MiB 35 GB/s,
// 12 x 64 MiB 39 GB/s (the link does 57 GB/s one way); VP8L_HOST_LANES / VP8L_HOST_CHUNK_MB /
// VP8L_HOST_DECODE_STREAMS override the defaults.
constexpr int kLanes = 16;  // at most
int env_int(const char *name, int fallback, int lo, int hi) {
  const char *v = getenv(name);
  if (!v) return fallback;
  const int x = atoi(v);
  return x < lo ? lo : x > hi ? hi : x;
}
const int g_lanes = env_int("VP8L_HOST_LANES", 8, 1, kLanes);
const size_t g_chunk_bytes = (size_t)env_int("VP8L_HOST_CHUNK_MB", 64, 1, 4096) << 20;  // pixel bytes per encode chunk (at least one image)
// Decode chunks (make_decode_chunks): twice the encode chunk size in pixels once a chunk holds a few MiB of streams or
// this many streams (small calls give a stream one workgroup: it takes hundreds to fill the device), 8 GiB of pixels at most.
const size_t g_decode_chunk_streams = (size_t)env_int("VP8L_HOST_DECODE_STREAMS", 256, 1, 1 << 20);
constexpr size_t kDecodeChunkMaxBytes = (size_t)8 << 30;

// Results handed to the caller live in page-locked host memory: a download into freshly allocated ordinary
// memory runs at 3-4 GB/s (first-touch page faults under a staged copy), into page-locked memory at the link's
// 50+ GB/s.  Page-locking is slow itself, so blocks given back through vp8l_free are kept for the next call
// (up to VP8L_HOST_POOL_MB, default 2048; 0 = plain malloc, nothing kept).
class HostPool {
 public:
  void *get(size_t bytes) {
    const size_t want = (std::max<size_t>(bytes, 1) + kGrain - 1) & ~(kGrain - 1);
    if (cap_ == 0) return malloc(want);
    {
      std::lock_guard<std::mutex> lock(mu_);
      auto it = idle_.lower_bound(want);
      if (it != idle_.end() && it->first <= 2 * want) {  // close enough in size
        void *p = it->second;
        idle_bytes_ -= it->first;
        live_[p] = it->first;
        idle_.erase(it);
        return p;
      }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, want, hipHostMallocPortable) != hipSuccess) {
      (void)hipGetLastError();
      return malloc(want);  // not tracked: put() says no and the caller frees it
    }
    std::lock_guard<std::mutex> lock(mu_);
    live_[p] = want;
    return p;
  }
  bool put(void *p) {
    size_t size = 0;
    {
      std::lock_guard<std::mutex> lock(mu_);
      auto it = live_.find(p);
      if (it == live_.end()) return false;
      size = it->second;
      live_.erase(it);
      if (idle_bytes_ + size <= cap_) {
        idle_.emplace(size, p);
        idle_bytes_ += size;
        return true;
      }
    }
    (void)hipHostFree(p);
    return true;
  }

 private:
  static constexpr size_t kGrain = 64 << 10;
  std::mutex mu_;
  std::map<void *, size_t> live_;
  std::multimap<size_t, void *> idle_;
  size_t idle_bytes_ = 0;
  const size_t cap_ = (size_t)env_int("VP8L_HOST_POOL_MB", 2048, 0, 1 << 20) << 20;
};
HostPool &host_pool() {
  static HostPool *p = new HostPool;  // intentionally leaked: HIP may be gone at exit
  return *p;
}

struct Lane {
  vp8l_encoder *enc = nullptr;
  vp8l_decoder *dec = nullptr;
  hipStream_t copy = nullptr;
  DeviceBuffer d_in, d_out;
};

struct Pipeline {
  std::mutex mu;  // one host-memory call at a time per pipeline
  int device = 0;
  Lane lanes[kLanes];
};

struct HostContexts {
  std::mutex mu;
  std::map<int, Pipeline *> enc;  // key: device * 256 + transform settings (callers with different flags never share option state)
  std::map<int, Pipeline *> dec;  // key: device
};
HostContexts &contexts() {
  static HostContexts *c = new HostContexts;  // intentionally leaked: HIP may be gone at exit
  return *c;
}

int get_pipeline(bool encoder, int device, int pred_bits, Pipeline **out) {
  int dev = device;
  int rc = select_device(device, &dev);
  if (rc != VP8L_OK) return rc;
  HostContexts &c = contexts();
  std::lock_guard<std::mutex> lock(c.mu);
  std::map<int, Pipeline *> &m = encoder ? c.enc : c.dec;
  const int key = encoder ? dev * (1 << 20) + pred_bits : dev;  // pred_bits: predictor bits | cross-colour bits << 4 | standard dialect << 8 | cache bits << 12 | entropy-image bits << 16
  auto it = m.find(key);
  if (it == m.end()) {
    Pipeline *p = new (std::nothrow) Pipeline;
    if (!p) return VP8L_ERR_NO_MEMORY;
    p->device = dev;
    it = m.emplace(key, p).first;
  }
  *out = it->second;
  return VP8L_OK;
}

// Creates what a lane needs on first use (the pipeline's mutex is held).
int lane_ready(Pipeline *p, Lane &l, bool encoder, int pred_bits) {
  if (!l.copy && !hip_ok(hipStreamCreateWithFlags(&l.copy, hipStreamNonBlocking), "hipStreamCreate")) return VP8L_ERR_HIP;
  if (encoder && !l.enc) {
    int rc = vp8l_encoder_create(p->device, &l.enc);
    if (rc != VP8L_OK) return rc;
    if (vp8l_encoder_set_option(l.enc, VP8L_OPT_PREDICTOR_BITS, pred_bits & 15) != VP8L_OK ||
        vp8l_encoder_set_option(l.enc, VP8L_OPT_CROSS_COLOR_BITS, (pred_bits >> 4) & 15) != VP8L_OK ||
        vp8l_encoder_set_option(l.enc, VP8L_OPT_STANDARD, (pred_bits >> 8) & 1) != VP8L_OK ||
        vp8l_encoder_set_option(l.enc, VP8L_OPT_CACHE_BITS, (pred_bits >> 12) & 15) != VP8L_OK ||
        vp8l_encoder_set_option(l.enc, VP8L_OPT_META_BITS, (pred_bits >> 16) & 15) != VP8L_OK)
      return VP8L_ERR_BAD_ARGUMENT;
  }
  if (!encoder && !l.dec) {
    int rc = vp8l_decoder_create(p->device, &l.dec);
    if (rc != VP8L_OK) return rc;
  }
  return VP8L_OK;
}

// Consecutive items [first[k], first[k+1]) form chunk k.  A chunk is closed before an item that would take it past
// `soft` bytes once it holds `min_items`, and in any case before one that would take it past `hard` bytes; it
// always holds at least one item.
std::vector<size_t> make_chunks(const std::vector<size_t> &bytes, size_t min_items, size_t soft, size_t hard) {
  std::vector<size_t> first(1, 0);
  size_t acc = 0, count = 0;
  for (size_t i = 0; i < bytes.size(); i++) {
    if (count && ((count >= min_items && acc + bytes[i] > soft) || acc + bytes[i] > hard)) { first.push_back(i); acc = 0; count = 0; }
    acc += bytes[i];
    count++;
  }
  first.push_back(bytes.size());
  return first;
}

// Chunks of a decode call.  A chunk should be small, so that its download overlaps the decoding of the next ones, and
// large enough for the decoder: the whole device works on any call with a few MiB of streams, however few they are, while
// small calls give a stream one workgroup and need hundreds of streams to fill the device.  So a chunk closes at
// `soft` bytes of pixels once it holds kDecodeChunkStreamBytes of streams or min_items streams, and at `hard` anyway.
// (128 4K streams: one chunk 8.4 GP/s, chunks of 128 MiB of pixels 10.8, of 64 MiB 8.4.)
constexpr size_t kDecodeChunkStreamBytes = (size_t)4 << 20;
std::vector<size_t> make_decode_chunks(const std::vector<size_t> &bytes, const std::vector<size_t> &stream_bytes, size_t min_items,
                                       size_t soft, size_t hard) {
  std::vector<size_t> first(1, 0);
  size_t acc = 0, count = 0, streams = 0;
  for (size_t i = 0; i < bytes.size(); i++) {
    const bool enough = count >= min_items || streams >= kDecodeChunkStreamBytes;
    if (count && ((enough && acc + bytes[i] > soft) || acc + bytes[i] > hard)) { first.push_back(i); acc = 0; count = 0; streams = 0; }
    acc += bytes[i];
    streams += stream_bytes[i];
    count++;
  }
  first.push_back(bytes.size());
  return first;
}

// Runs work(lane, chunk) for every chunk on up to kLanes host threads; returns the first failure.
template <class Work>
int run_chunks(Pipeline *p, size_t nchunks, Work work) {
  std::atomic<size_t> next(0);
  std::atomic<int> failure(VP8L_OK);
  std::string failure_text;
  std::mutex text_mu;
  auto body = [&](int lane) {
    if (hipSetDevice(p->device) != hipSuccess) { failure = VP8L_ERR_HIP; return; }
    for (;;) {
      const size_t c = next.fetch_add(1);
      if (c >= nchunks || failure.load() != VP8L_OK) return;
      const int rc = work(p->lanes[lane], c);
      if (rc != VP8L_OK) {
        (void)hipStreamSynchronize(p->lanes[lane].copy);  // no copy may still write into buffers the caller is about to free
        std::lock_guard<std::mutex> lock(text_mu);
        if (failure.load() == VP8L_OK) { failure = rc; failure_text = last_error_slot(); }  // the text is per thread: carry it over
        return;
      }
    }
  };
  const int nthreads = (int)std::min<size_t>(g_lanes, nchunks);
  std::vector<std::thread> threads;
  for (int t = 1; t < nthreads; t++) threads.emplace_back(body, t);
  body(0);
  for (auto &t : threads) t.join();
  if (failure.load() != VP8L_OK) last_error_slot() = failure_text;
  return failure.load();
}

}  // namespace
}  // namespace vp8l

using namespace vp8l;

extern "C" int vp8l_init(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    last_error_slot() = "no HIP device available";
    return VP8L_ERR_NO_DEVICE;
  }
  int usable = 0;
  for (int d = 0; d < count; d++) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) == hipSuccess && std::string(prop.gcnArchName).rfind("gfx950", 0) == 0) usable++;
  }
  if (usable > 0) noalloc_setup_current_device();
  return usable > 0 ? usable : VP8L_ERR_NO_DEVICE;
}

extern "C" const char *vp8l_status_string(int status) {
  switch (status) {  // messages of the matching reference exceptions (exception.h:14-153) where they have one
    case VP8L_OK: return "ok";
    case VP8L_ERR_INVALID_ARGB: return "Invalid ARGB image";
    case VP8L_ERR_TOO_BIG_ARGB: return "Too big ARGB Image. Max allowed width(height)=16384";
    case VP8L_ERR_TOO_BIG_CODE_LENGTH: return "Too big Huffman code length";
    case VP8L_ERR_UNSUPPORTED_VP8: return "VP8 is unsupported";
    case VP8L_ERR_INVALID_VP8L: return "Invalid VP8L";
    case VP8L_ERR_INVALID_HUFFMAN: return "Invalid format";
    case VP8L_ERR_NO_DEVICE: return "no usable gfx950 device";
    case VP8L_ERR_NO_MEMORY: return "Memory allocation exception";
    case VP8L_ERR_HIP: return "HIP runtime error";
    case VP8L_ERR_BAD_ARGUMENT: return "bad argument";
    case VP8L_ERR_WORKSPACE: return "decode workspace too small";
  }
  return "unknown status";
}

extern "C" const char *vp8l_last_error(void) { return last_error_slot().c_str(); }

extern "C" void vp8l_free(void *p) {
  if (p && !host_pool().put(p)) free(p);
}

extern "C" int vp8l_get_info(const uint8_t *data, size_t len, uint32_t *width, uint32_t *height) {
  if (!data || !width || !height) return VP8L_ERR_BAD_ARGUMENT;
  uint8_t b[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // bytes past the end read as zero (bit_readed.h:27-42)
  memcpy(b, data, len < 9 ? len : 9);
  if (b[4] != 0x2f) return VP8L_ERR_UNSUPPORTED_VP8;  // vp8l.h:86-88
  const uint32_t bits = (uint32_t)b[5] | ((uint32_t)b[6] << 8) | ((uint32_t)b[7] << 16) | ((uint32_t)b[8] << 24);
  *width = (bits & 0x3fff) + 1;
  *height = ((bits >> 14) & 0x3fff) + 1;
  return VP8L_OK;
}

namespace {
int encode_batch_with(int device, int pred_bits, size_t n, const uint32_t *const *argb, const uint32_t *width,
                      const uint32_t *height, uint8_t **out, size_t *out_len, int *status) {
  if (!argb || !width || !height || !out || !out_len || !status) return VP8L_ERR_BAD_ARGUMENT;
  if (n == 0) return VP8L_OK;
  Pipeline *p = nullptr;
  int rc = get_pipeline(true, device, pred_bits, &p);
  if (rc != VP8L_OK) return rc;
  std::lock_guard<std::mutex> lock(p->mu);
  std::vector<size_t> bytes(n, 0);
  for (size_t i = 0; i < n; i++) {
    out[i] = nullptr;
    out_len[i] = 0;
    status[i] = VP8L_OK;
    const bool valid = argb[i] && width[i] && height[i] && width[i] <= kMaxDim && height[i] <= kMaxDim;
    if (valid) bytes[i] = (size_t)width[i] * height[i] * 4;
  }
  const std::vector<size_t> first = make_chunks(bytes, 1, g_chunk_bytes, g_chunk_bytes);
  for (int l = 0; l < g_lanes && l < (int)first.size() - 1; l++)
    if ((rc = lane_ready(p, p->lanes[l], true, pred_bits)) != VP8L_OK) return rc;
  auto work = [&](Lane &lane, size_t c) -> int {
    const size_t i0 = first[c], m = first[c + 1] - i0;
    std::vector<size_t> in_off(m, 0);
    size_t in_total = 0, out_total = 0;
    for (size_t k = 0; k < m; k++) {
      in_off[k] = in_total;
      in_total += (bytes[i0 + k] + 255) & ~(size_t)255;
      if (bytes[i0 + k]) out_total += vp8l_encode_bound(width[i0 + k], height[i0 + k]);
    }
    if (!lane.d_in.reserve(in_total + 256) || !lane.d_out.reserve(out_total + 256)) return VP8L_ERR_NO_MEMORY;
    std::vector<const uint32_t *> d_ptrs(m, nullptr);
    for (size_t k = 0; k < m; k++) {
      uint8_t *dst = lane.d_in.as<uint8_t>() + in_off[k];
      // an invalid image keeps a non-null pointer only if the caller gave one, so the device call reports it the same way
      d_ptrs[k] = argb[i0 + k] ? reinterpret_cast<const uint32_t *>(dst) : nullptr;
      if (bytes[i0 + k] && !hip_ok(hipMemcpyAsync(dst, argb[i0 + k], bytes[i0 + k], hipMemcpyHostToDevice, lane.copy), "upload image"))
        return VP8L_ERR_HIP;
    }
    if (!hip_ok(hipStreamSynchronize(lane.copy), "upload images")) return VP8L_ERR_HIP;
    std::vector<size_t> off(m, 0);
    const int r = vp8l_encoder_encode_device(lane.enc, m, d_ptrs.data(), width + i0, height + i0, lane.d_out.as<uint8_t>(), out_total + 256,
                                             off.data(), out_len + i0, status + i0);
    if (r != VP8L_OK) return r;
    for (size_t k = 0; k < m; k++) {
      const size_t i = i0 + k;
      if (status[i] != VP8L_OK) { out_len[i] = 0; continue; }
      out[i] = static_cast<uint8_t *>(host_pool().get(out_len[i]));
      if (!out[i]) { status[i] = VP8L_ERR_NO_MEMORY; out_len[i] = 0; continue; }
      if (!hip_ok(hipMemcpyAsync(out[i], lane.d_out.as<uint8_t>() + off[k], out_len[i], hipMemcpyDeviceToHost, lane.copy), "download stream"))
        return VP8L_ERR_HIP;
    }
    if (!hip_ok(hipStreamSynchronize(lane.copy), "download streams")) return VP8L_ERR_HIP;
    return VP8L_OK;
  };
  rc = run_chunks(p, first.size() - 1, work);
  if (rc != VP8L_OK) {
    for (size_t i = 0; i < n; i++) { vp8l_free(out[i]); out[i] = nullptr; out_len[i] = 0; }
  }
  return rc;
}
}  // namespace

extern "C" int vp8l_encode_batch(int device, size_t n, const uint32_t *const *argb, const uint32_t *width,
                                 const uint32_t *height, uint8_t **out, size_t *out_len, int *status) {
  return guarded([&] { return encode_batch_with(device, 0, n, argb, width, height, out, out_len, status); });
}

extern "C" int vp8l_encode(const uint32_t *argb, uint32_t width, uint32_t height, uint8_t **out, size_t *out_len, uint32_t flags) {
  const int pred_bits = (int)(flags & VP8L_FLAG_PREDICTOR_BITS_MASK), cross_bits = (int)((flags & VP8L_FLAG_CROSS_COLOR_BITS_MASK) >> 4);
  if (!out || !out_len || (flags & ~(uint32_t)(VP8L_FLAG_PREDICTOR_BITS_MASK | VP8L_FLAG_CROSS_COLOR_BITS_MASK | VP8L_FLAG_STANDARD | VP8L_FLAG_CACHE_BITS_MASK | VP8L_FLAG_META_BITS_MASK)) || pred_bits == 1 || pred_bits > 9 ||
      cross_bits == 1 || cross_bits > 9 || ((flags >> 12) & 15) > 11 || ((flags >> 16) & 15) == 1 || ((flags >> 16) & 15) > 9)
    return VP8L_ERR_BAD_ARGUMENT;
  // argument checks of the reference constructor, in its order (vp8l.h:587-590)
  if (!argb || width == 0 || height == 0) return VP8L_ERR_INVALID_ARGB;
  if (width > kMaxDim || height > kMaxDim) return VP8L_ERR_TOO_BIG_ARGB;
  int status = VP8L_OK;
  const int rc = guarded([&] { return encode_batch_with(-1, pred_bits | (cross_bits << 4) | ((flags & VP8L_FLAG_STANDARD) ? 256 : 0) | (int)(flags & (VP8L_FLAG_CACHE_BITS_MASK | VP8L_FLAG_META_BITS_MASK)), 1, &argb, &width, &height, out, out_len, &status); });
  return rc != VP8L_OK ? rc : status;
}

namespace {
int decode_batch_impl(int device, size_t n, const uint8_t *const *data, const size_t *len, uint32_t **argb,
                      uint32_t *width, uint32_t *height, int *status) {
  if (!data || !len || !argb || !width || !height || !status) return VP8L_ERR_BAD_ARGUMENT;
  if (n == 0) return VP8L_OK;
  Pipeline *p = nullptr;
  int rc = get_pipeline(false, device, 0, &p);
  if (rc != VP8L_OK) return rc;
  std::lock_guard<std::mutex> lock(p->mu);
  // streams that already fail here keep their status; the device sees only the others
  std::vector<size_t> idx, bytes, stream_bytes;
  for (size_t i = 0; i < n; i++) {
    argb[i] = nullptr;
    width[i] = height[i] = 0;
    status[i] = data[i] ? vp8l_get_info(data[i], len[i], &width[i], &height[i]) : VP8L_ERR_BAD_ARGUMENT;
    if (status[i] != VP8L_OK) continue;
    idx.push_back(i);
    bytes.push_back((size_t)width[i] * height[i] * 4);
    stream_bytes.push_back(len[i]);
  }
  if (idx.empty()) return VP8L_OK;
  const std::vector<size_t> first = make_decode_chunks(bytes, stream_bytes, g_decode_chunk_streams, 2 * g_chunk_bytes, kDecodeChunkMaxBytes);
  for (int l = 0; l < g_lanes && l < (int)first.size() - 1; l++)
    if ((rc = lane_ready(p, p->lanes[l], false, 0)) != VP8L_OK) return rc;
  auto work = [&](Lane &lane, size_t c) -> int {
    const size_t k0 = first[c], m = first[c + 1] - k0;
    std::vector<size_t> in_off(m, 0), out_off(m, 0), v_len(m, 0);
    std::vector<uint32_t> v_w(m), v_h(m);
    size_t in_total = 0, out_total = 0;
    for (size_t k = 0; k < m; k++) {
      const size_t i = idx[k0 + k];
      in_off[k] = in_total;
      in_total += (len[i] + 255 + 8) & ~(size_t)255;
      out_off[k] = out_total;
      out_total += (bytes[k0 + k] + 255) & ~(size_t)255;
      v_len[k] = len[i];
      v_w[k] = width[i];
      v_h[k] = height[i];
    }
    if (!lane.d_in.reserve(in_total + 256) || !lane.d_out.reserve(out_total + 256)) return VP8L_ERR_NO_MEMORY;
    std::vector<const uint8_t *> d_data(m, nullptr);
    std::vector<uint32_t *> d_argb(m, nullptr);
    for (size_t k = 0; k < m; k++) {
      const size_t i = idx[k0 + k];
      uint8_t *dst = lane.d_in.as<uint8_t>() + in_off[k];
      if (len[i] && !hip_ok(hipMemcpyAsync(dst, data[i], len[i], hipMemcpyHostToDevice, lane.copy), "upload stream")) return VP8L_ERR_HIP;
      d_data[k] = dst;
      d_argb[k] = reinterpret_cast<uint32_t *>(lane.d_out.as<uint8_t>() + out_off[k]);
    }
    if (!hip_ok(hipStreamSynchronize(lane.copy), "upload streams")) return VP8L_ERR_HIP;
    std::vector<int> v_status(m, VP8L_OK);
    const int r = vp8l_decoder_decode_device(lane.dec, m, d_data.data(), v_len.data(), v_w.data(), v_h.data(), d_argb.data(), v_status.data());
    if (r != VP8L_OK) return r;
    for (size_t k = 0; k < m; k++) {
      const size_t i = idx[k0 + k];
      status[i] = v_status[k];
      if (status[i] != VP8L_OK) continue;
      argb[i] = static_cast<uint32_t *>(host_pool().get(bytes[k0 + k]));
      if (!argb[i]) { status[i] = VP8L_ERR_NO_MEMORY; continue; }
      if (!hip_ok(hipMemcpyAsync(argb[i], d_argb[k], bytes[k0 + k], hipMemcpyDeviceToHost, lane.copy), "download image")) return VP8L_ERR_HIP;
    }
    if (!hip_ok(hipStreamSynchronize(lane.copy), "download images")) return VP8L_ERR_HIP;
    return VP8L_OK;
  };
  rc = run_chunks(p, first.size() - 1, work);
  if (rc != VP8L_OK) {
    for (size_t i = 0; i < n; i++) { vp8l_free(argb[i]); argb[i] = nullptr; }
  }
  return rc;
}

}  // namespace

extern "C" int vp8l_decode_batch(int device, size_t n, const uint8_t *const *data, const size_t *len, uint32_t **argb,
                                 uint32_t *width, uint32_t *height, int *status) {
  return guarded([&] { return decode_batch_impl(device, n, data, len, argb, width, height, status); });
}

extern "C" int vp8l_decode(const uint8_t *data, size_t len, uint32_t **argb, uint32_t *width, uint32_t *height) {
  if (!data || !argb || !width || !height) return VP8L_ERR_BAD_ARGUMENT;
  int status = VP8L_OK;
  const int rc = vp8l_decode_batch(-1, 1, &data, &len, argb, width, height, &status);
  return rc != VP8L_OK ? rc : status;
}

// ---- batches dealt out to several GPUs ------------------------------------------------------------
namespace {
// Runs part(device, first_item, count) for every device's block on a host thread of its own; the first failure wins.
template <class Part>
int run_on_devices(const int *devices, size_t ndevices, const std::vector<uint64_t> &weight, Part part) {
  if (!devices || ndevices == 0) return VP8L_ERR_BAD_ARGUMENT;
  const std::vector<size_t> first = split_by_weight(weight, ndevices);
  std::vector<int> rcs(ndevices, VP8L_OK);
  std::vector<std::string> texts(ndevices);
  std::vector<std::thread> threads;
  auto body = [&](size_t d) {
    rcs[d] = guarded([&] { return first[d + 1] > first[d] ? part(devices[d], first[d], first[d + 1] - first[d]) : (int)VP8L_OK; });
    if (rcs[d] != VP8L_OK) texts[d] = last_error_slot();
  };
  // a thread that cannot be started (std::system_error) must not unwind past joinable threads: its part runs here instead
  std::vector<size_t> inline_parts;
  threads.reserve(ndevices);
  for (size_t d = 1; d < ndevices; d++) {
    try {
      threads.emplace_back(body, d);
    } catch (const std::exception &) {
      inline_parts.push_back(d);
    }
  }
  body(0);
  for (size_t d : inline_parts) body(d);
  for (auto &t : threads) t.join();
  for (size_t d = 0; d < ndevices; d++)
    if (rcs[d] != VP8L_OK) { last_error_slot() = texts[d]; return rcs[d]; }
  return VP8L_OK;
}
}  // namespace

extern "C" int vp8l_encode_batch_devices(const int *devices, size_t ndevices, size_t n, const uint32_t *const *argb, const uint32_t *width,
                                         const uint32_t *height, uint8_t **out, size_t *out_len, int *status) {
  if (!argb || !width || !height || !out || !out_len || !status) return VP8L_ERR_BAD_ARGUMENT;
  return guarded([&] {
    std::vector<uint64_t> weight(n);
    for (size_t i = 0; i < n; i++) weight[i] = (uint64_t)width[i] * height[i];
    return run_on_devices(devices, ndevices, weight, [&](int device, size_t i0, size_t m) {
      return encode_batch_with(device, 0, m, argb + i0, width + i0, height + i0, out + i0, out_len + i0, status + i0);
    });
  });
}

extern "C" int vp8l_decode_batch_devices(const int *devices, size_t ndevices, size_t n, const uint8_t *const *data, const size_t *len,
                                         uint32_t **argb, uint32_t *width, uint32_t *height, int *status) {
  if (!data || !len || !argb || !width || !height || !status) return VP8L_ERR_BAD_ARGUMENT;
  return guarded([&] {
    std::vector<uint64_t> weight(n, 1);
    for (size_t i = 0; i < n; i++) {
      uint32_t w = 0, h = 0;
      if (data[i] && vp8l_get_info(data[i], len[i], &w, &h) == VP8L_OK) weight[i] = (uint64_t)w * h;
    }
    return run_on_devices(devices, ndevices, weight, [&](int device, size_t i0, size_t m) {
      return decode_batch_impl(device, m, data + i0, len + i0, argb + i0, width + i0, height + i0, status + i0);
    });
  });
}
