// vp8l_capi.hip -- host-memory entry points of the C ABI (include/vp8l_amd.h) and shared host
// helpers.  The host-memory forms stage images through HBM and call the device-memory batch
// forms; all codec work happens in the kernels of vp8l_encode.hip / vp8l_decode.hip.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "vp8l_amd.h"
#include "vp8l_common.h"
#include "vp8l_host.h"

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

// One shared context pair per device for the host-memory entry points.
struct HostContexts {
  std::mutex mu;
  std::map<int, vp8l_encoder *> enc;
  std::map<int, vp8l_decoder *> dec;
};
HostContexts &contexts() {
  static HostContexts *c = new HostContexts;  // intentionally leaked: HIP may be gone at exit
  return *c;
}

// one shared context per (device, predictor setting): callers with different flags never share option state
int get_encoder(int device, int pred_bits, vp8l_encoder **e) {
  int dev = device;
  int rc = select_device(device, &dev);
  if (rc != VP8L_OK) return rc;
  HostContexts &c = contexts();
  std::lock_guard<std::mutex> lock(c.mu);
  const int key = dev * 16 + pred_bits;
  auto it = c.enc.find(key);
  if (it == c.enc.end()) {
    vp8l_encoder *ne = nullptr;
    rc = vp8l_encoder_create(dev, &ne);
    if (rc != VP8L_OK) return rc;
    if (pred_bits && vp8l_encoder_set_option(ne, VP8L_OPT_PREDICTOR_BITS, pred_bits) != VP8L_OK) {
      vp8l_encoder_destroy(ne);
      return VP8L_ERR_BAD_ARGUMENT;
    }
    it = c.enc.emplace(key, ne).first;
  }
  *e = it->second;
  return VP8L_OK;
}

int get_decoder(int device, vp8l_decoder **d) {
  int dev = device;
  int rc = select_device(device, &dev);
  if (rc != VP8L_OK) return rc;
  HostContexts &c = contexts();
  std::lock_guard<std::mutex> lock(c.mu);
  auto it = c.dec.find(dev);
  if (it == c.dec.end()) {
    vp8l_decoder *nd = nullptr;
    rc = vp8l_decoder_create(dev, &nd);
    if (rc != VP8L_OK) return rc;
    it = c.dec.emplace(dev, nd).first;
  }
  *d = it->second;
  return VP8L_OK;
}

struct DevMem {
  void *p = nullptr;
  ~DevMem() { if (p) (void)hipFree(p); }
  bool alloc(size_t bytes) { return hip_ok(hipMalloc(&p, bytes ? bytes : 4), "hipMalloc"); }
};

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

extern "C" void vp8l_free(void *p) { free(p); }

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
  vp8l_encoder *e = nullptr;
  int rc = get_encoder(device, pred_bits, &e);
  if (rc != VP8L_OK) return rc;
  // stage inputs in one device buffer, outputs in another
  std::vector<size_t> in_off(n, 0);
  std::vector<const uint32_t *> d_ptrs(n, nullptr);
  size_t in_total = 0, out_total = 0;
  for (size_t i = 0; i < n; i++) {
    out[i] = nullptr;
    out_len[i] = 0;
    const bool valid = argb[i] && width[i] && height[i] && width[i] <= kMaxDim && height[i] <= kMaxDim;
    if (!valid) continue;
    in_off[i] = in_total;
    in_total += (((size_t)width[i] * height[i] * 4) + 255) & ~(size_t)255;
    out_total += vp8l_encode_bound(width[i], height[i]);
  }
  DevMem d_in, d_out;
  if (!d_in.alloc(in_total) || !d_out.alloc(out_total)) return VP8L_ERR_NO_MEMORY;
  for (size_t i = 0; i < n; i++) {
    const bool valid = argb[i] && width[i] && height[i] && width[i] <= kMaxDim && height[i] <= kMaxDim;
    if (!valid) { d_ptrs[i] = argb[i] ? reinterpret_cast<const uint32_t *>(d_in.p) : nullptr; continue; }
    uint8_t *dst = static_cast<uint8_t *>(d_in.p) + in_off[i];
    if (!hip_ok(hipMemcpy(dst, argb[i], (size_t)width[i] * height[i] * 4, hipMemcpyHostToDevice), "upload image")) return VP8L_ERR_HIP;
    d_ptrs[i] = reinterpret_cast<const uint32_t *>(dst);
  }
  std::vector<size_t> off(n, 0);
  rc = vp8l_encoder_encode_device(e, n, d_ptrs.data(), width, height, static_cast<uint8_t *>(d_out.p), out_total, off.data(), out_len, status);
  if (rc != VP8L_OK) return rc;
  for (size_t i = 0; i < n; i++) {
    if (status[i] != VP8L_OK) { out_len[i] = 0; continue; }
    out[i] = static_cast<uint8_t *>(malloc(out_len[i] ? out_len[i] : 1));
    if (!out[i]) { status[i] = VP8L_ERR_NO_MEMORY; continue; }
    if (!hip_ok(hipMemcpy(out[i], static_cast<uint8_t *>(d_out.p) + off[i], out_len[i], hipMemcpyDeviceToHost), "download stream")) {
      free(out[i]);
      out[i] = nullptr;
      return VP8L_ERR_HIP;
    }
  }
  return VP8L_OK;
}
}  // namespace

extern "C" int vp8l_encode_batch(int device, size_t n, const uint32_t *const *argb, const uint32_t *width,
                                 const uint32_t *height, uint8_t **out, size_t *out_len, int *status) {
  return encode_batch_with(device, 0, n, argb, width, height, out, out_len, status);
}

extern "C" int vp8l_encode(const uint32_t *argb, uint32_t width, uint32_t height, uint8_t **out, size_t *out_len, uint32_t flags) {
  const int pred_bits = (int)(flags & VP8L_FLAG_PREDICTOR_BITS_MASK);
  if (!out || !out_len || (flags & ~(uint32_t)VP8L_FLAG_PREDICTOR_BITS_MASK) || pred_bits == 1 || pred_bits > 9) return VP8L_ERR_BAD_ARGUMENT;
  // argument checks of the reference constructor, in its order (vp8l.h:587-590)
  if (!argb || width == 0 || height == 0) return VP8L_ERR_INVALID_ARGB;
  if (width > kMaxDim || height > kMaxDim) return VP8L_ERR_TOO_BIG_ARGB;
  int status = VP8L_OK;
  const int rc = encode_batch_with(-1, pred_bits, 1, &argb, &width, &height, out, out_len, &status);
  return rc != VP8L_OK ? rc : status;
}

extern "C" int vp8l_decode_batch(int device, size_t n, const uint8_t *const *data, const size_t *len, uint32_t **argb,
                                 uint32_t *width, uint32_t *height, int *status) {
  if (!data || !len || !argb || !width || !height || !status) return VP8L_ERR_BAD_ARGUMENT;
  if (n == 0) return VP8L_OK;
  vp8l_decoder *d = nullptr;
  int rc = get_decoder(device, &d);
  if (rc != VP8L_OK) return rc;
  std::vector<size_t> in_off(n, 0), out_off(n, 0);
  size_t in_total = 0, out_total = 0;
  std::vector<char> valid(n, 0);
  for (size_t i = 0; i < n; i++) {
    argb[i] = nullptr;
    width[i] = height[i] = 0;
    status[i] = data[i] ? vp8l_get_info(data[i], len[i], &width[i], &height[i]) : VP8L_ERR_BAD_ARGUMENT;
    if (status[i] != VP8L_OK) continue;
    valid[i] = 1;
    in_off[i] = in_total;
    in_total += (len[i] + 255 + 8) & ~(size_t)255;
    out_off[i] = out_total;
    out_total += (((size_t)width[i] * height[i] * 4) + 255) & ~(size_t)255;
  }
  DevMem d_in, d_out;
  if (!d_in.alloc(in_total) || !d_out.alloc(out_total)) return VP8L_ERR_NO_MEMORY;
  std::vector<const uint8_t *> d_data(n, nullptr);
  std::vector<uint32_t *> d_argb(n, nullptr);
  for (size_t i = 0; i < n; i++) {
    if (!valid[i]) continue;
    uint8_t *dst = static_cast<uint8_t *>(d_in.p) + in_off[i];
    if (len[i] && !hip_ok(hipMemcpy(dst, data[i], len[i], hipMemcpyHostToDevice), "upload stream")) return VP8L_ERR_HIP;
    d_data[i] = dst;
    d_argb[i] = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(d_out.p) + out_off[i]);
  }
  // images that already failed keep their status: give the device call only the valid ones
  std::vector<const uint8_t *> v_data;
  std::vector<size_t> v_len, v_idx;
  std::vector<uint32_t> v_w, v_h;
  std::vector<uint32_t *> v_out;
  for (size_t i = 0; i < n; i++)
    if (valid[i]) { v_idx.push_back(i); v_data.push_back(d_data[i]); v_len.push_back(len[i]); v_w.push_back(width[i]); v_h.push_back(height[i]); v_out.push_back(d_argb[i]); }
  std::vector<int> v_status(v_idx.size(), VP8L_OK);
  if (!v_idx.empty()) {
    rc = vp8l_decoder_decode_device(d, v_idx.size(), v_data.data(), v_len.data(), v_w.data(), v_h.data(), v_out.data(), v_status.data());
    if (rc != VP8L_OK) return rc;
  }
  for (size_t k = 0; k < v_idx.size(); k++) {
    const size_t i = v_idx[k];
    status[i] = v_status[k];
    if (status[i] != VP8L_OK) continue;
    const size_t bytes = (size_t)width[i] * height[i] * 4;
    argb[i] = static_cast<uint32_t *>(malloc(bytes));
    if (!argb[i]) { status[i] = VP8L_ERR_NO_MEMORY; continue; }
    if (!hip_ok(hipMemcpy(argb[i], d_argb[i], bytes, hipMemcpyDeviceToHost), "download image")) {
      free(argb[i]);
      argb[i] = nullptr;
      return VP8L_ERR_HIP;
    }
  }
  return VP8L_OK;
}

extern "C" int vp8l_decode(const uint8_t *data, size_t len, uint32_t **argb, uint32_t *width, uint32_t *height) {
  if (!data || !argb || !width || !height) return VP8L_ERR_BAD_ARGUMENT;
  int status = VP8L_OK;
  const int rc = vp8l_decode_batch(-1, 1, &data, &len, argb, width, height, &status);
  return rc != VP8L_OK ? rc : status;
}
