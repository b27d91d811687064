// vp8l_host.h -- small host-side helpers shared by the encoder, decoder and C ABI sources.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <exception>
#include <new>
#include <string>

#include "vp8l_amd.h"

namespace vp8l {

std::string &last_error_slot();

// Records the failure text for vp8l_last_error() and returns false on any HIP error.
bool hip_ok(hipError_t e, const char *what);

// device < 0 keeps the current device.  Fails with VP8L_ERR_NO_DEVICE when no gfx950 is usable.
int select_device(int device, int *chosen);

// No C++ exception may cross the C boundary: allocation failures of the host-side containers and threads become
// VP8L_ERR_NO_MEMORY, anything else VP8L_ERR_HIP with its text in vp8l_last_error().
template <class Body>
int guarded(Body body) {
  try {
    return body();
  } catch (const std::bad_alloc &) {
    return VP8L_ERR_NO_MEMORY;
  } catch (const std::exception &e) {
    try { last_error_slot() = std::string("host exception: ") + e.what(); } catch (...) {}
    return VP8L_ERR_HIP;
  } catch (...) {
    return VP8L_ERR_HIP;
  }
}

// Looks up the address of the static no-allocation workspace on every device (vp8l_encode.hip).
void noalloc_setup_current_device();

// Grow-only device allocation (the contexts keep their workspace between calls).
struct DeviceBuffer {
  void *ptr = nullptr;
  size_t cap = 0;
  bool reserve(size_t bytes) {
    if (bytes <= cap) return true;
    release();
    const size_t want = bytes + bytes / 8 + 4096;
    if (!hip_ok(hipMalloc(&ptr, want), "hipMalloc")) { ptr = nullptr; return false; }
    cap = want;
    return true;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
  }
  template <class T>
  T *as() const { return static_cast<T *>(ptr); }
};

}  // namespace vp8l
