// ref_driver_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// extern "C" shims over the REFERENCE's own encoder and decoder drivers
// (webp/vp8l/vp8l.h:585-615 and :330-351), compiled together with the reference's own .cpp
// files where they lie (oracle/Makefile, target `ref`).  vp8l.h includes <png.h> without using
// it; the Makefile builds this file only where a real libpng header is installed (never a
// stand-in).  Nothing here restates an algorithm: each function moves data in and out of a
// reference object.  The reference's decoder checks neither back references nor palette rows
// (vp8l.h:298-300, transform.h:70-86), so every decode runs in a forked child: a stream that
// makes the reference crash yields REF_CRASHED instead of taking the test process down.
#include "webp/platform.h"
#include "webp/exception/exception.h"
#include "webp/utils/utils.h"
#include "webp/vp8l/vp8l.h"

#include <fcntl.h>
#include <signal.h>
#include <sys/resource.h>
#include <sys/wait.h>
#include <unistd.h>

using namespace webp;

namespace {
struct Init {
  Init() { vp8l::huffman_io::init_array(); }  // webp.cpp:128
} g_init;

enum {
  REF_OK = 0,
  REF_INVALID_ARGB = -1,
  REF_TOO_BIG_ARGB = -2,
  REF_TOO_BIG_CODE = -3,
  REF_UNSUPPORTED_VP8 = -4,
  REF_INVALID_VP8L = -5,
  REF_INVALID_HUFFMAN = -6,
  REF_OTHER_EXCEPTION = -7,
  REF_NO_ROOM = -8,
  REF_CRASHED = -99,
};

// The encoder reports progress with printf (vp8l.h:591,608,611) and the exception constructors
// write to std::cout: fd 1 points at /dev/null while a reference object is alive.
struct Silence {
  int saved;
  Silence() {
    fflush(stdout);
    std::cout.flush();
    saved = dup(1);
    int nul = open("/dev/null", O_WRONLY);
    if (nul >= 0) {
      dup2(nul, 1);
      close(nul);
    }
  }
  ~Silence() {
    fflush(stdout);
    std::cout.flush();
    if (saved >= 0) {
      dup2(saved, 1);
      close(saved);
    }
  }
};

size_t writer_bytes(const utils::BitWriter &bw, uint8_t *buf, size_t cap) {
  char path[] = "/tmp/ref_driver_XXXXXX";
  int fd = mkstemp(path);
  if (fd < 0) return 0;
  close(fd);
  unlink(path);  // save2file appends (bit_writer.h:65-90)
  bw.save2file(path);
  FILE *fp = fopen(path, "rb");
  size_t n = 0;
  if (fp) {
    n = fread(buf, 1, cap, fp);
    fclose(fp);
  }
  unlink(path);
  return n;
}

bool write_all(int fd, const void *p, size_t n) {
  const char *c = (const char *)p;
  while (n) {
    ssize_t k = write(fd, c, n);
    if (k <= 0) return false;
    c += k;
    n -= (size_t)k;
  }
  return true;
}

bool read_all(int fd, void *p, size_t n) {
  char *c = (char *)p;
  while (n) {
    ssize_t k = read(fd, c, n);
    if (k <= 0) return false;
    c += k;
    n -= (size_t)k;
  }
  return true;
}
}  // namespace

extern "C" {

// vp8l.h:585-615: VP8_LOSSLESS_ENCODER(argb, width, height) then get_bit_writer().
// Returns the stream's byte count (BitWriter::size()), or a negative REF_* status.
long ref_encode(const uint32_t *argb, uint32_t npixels, uint32_t w, uint32_t h, uint8_t *out, size_t cap) {
  Silence quiet;
  try {
    utils::pixel_array img(npixels);
    for (uint32_t i = 0; i < npixels; i++) img[i] = argb[i];
    size_t ww = w, hh = h;
    vp8l::VP8_LOSSLESS_ENCODER enc(img, ww, hh);
    size_t n = enc.get_bit_writer().size();
    if (n > cap) return REF_NO_ROOM;
    size_t got = writer_bytes(enc.get_bit_writer(), out, cap);
    return got == n ? (long)n : (long)REF_OTHER_EXCEPTION;
  } catch (exception::InvalidARGBImage &) {
    return REF_INVALID_ARGB;
  } catch (exception::TooBigARGBImage &) {
    return REF_TOO_BIG_ARGB;
  } catch (exception::TooBigCodeLength &) {
    return REF_TOO_BIG_CODE;
  } catch (exception::Exception &) {
    return REF_OTHER_EXCEPTION;
  }
}

// vp8l.h:330-351: VP8_LOSSLESS_DECODER(data, len, argb_out), image_width(), image_height().
// Runs in a forked child (see the header comment).  Returns REF_OK and fills argb (up to
// cap_pixels), *w, *h; or the status of the exception the reference threw; or REF_CRASHED.
int ref_decode(const uint8_t *data, uint32_t len, uint32_t *argb, size_t cap_pixels, uint32_t *w, uint32_t *h) {
  int fds[2];
  if (pipe(fds) != 0) return REF_OTHER_EXCEPTION;
  fflush(stdout);
  fflush(stderr);
  pid_t pid = fork();
  if (pid < 0) {
    close(fds[0]);
    close(fds[1]);
    return REF_OTHER_EXCEPTION;
  }
  if (pid == 0) {
    close(fds[0]);
    struct rlimit lim;
    lim.rlim_cur = lim.rlim_max = 20;  // CPU seconds: a damaged stream must not spin forever
    setrlimit(RLIMIT_CPU, &lim);
    lim.rlim_cur = lim.rlim_max = 0;
    setrlimit(RLIMIT_CORE, &lim);
    int nul = open("/dev/null", O_WRONLY);
    if (nul >= 0) {
      dup2(nul, 1);
      dup2(nul, 2);
    }
    // the parent may have fault handlers that write a report (Python's faulthandler): die quietly
    const int fatal[] = {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL, SIGXCPU};
    for (size_t i = 0; i < sizeof fatal / sizeof fatal[0]; i++) signal(fatal[i], SIG_DFL);
    int32_t head[3] = {REF_OK, 0, 0};
    // slack: the reference reader touches a byte past the end before it reports the end
    // (bit_readed.h:39)
    uint8_t *copy = new uint8_t[(size_t)len + 64];
    memcpy(copy, data, len);
    memset(copy + len, 0, 64);
    try {
      utils::pixel_array img;
      vp8l::VP8_LOSSLESS_DECODER dec(copy, len, img);
      head[1] = (int32_t)dec.image_width();
      head[2] = (int32_t)dec.image_height();
      size_t n = (size_t)head[1] * (size_t)head[2];
      write_all(fds[1], head, sizeof head);
      write_all(fds[1], &img[0], 4 * n);
    } catch (exception::UnsupportedVP8 &) {
      head[0] = REF_UNSUPPORTED_VP8;
      write_all(fds[1], head, sizeof head);
    } catch (exception::InvalidVP8L &) {
      head[0] = REF_INVALID_VP8L;
      write_all(fds[1], head, sizeof head);
    } catch (exception::InvalidHuffman &) {
      head[0] = REF_INVALID_HUFFMAN;
      write_all(fds[1], head, sizeof head);
    } catch (exception::Exception &) {
      head[0] = REF_OTHER_EXCEPTION;
      write_all(fds[1], head, sizeof head);
    } catch (...) {
      head[0] = REF_OTHER_EXCEPTION;
      write_all(fds[1], head, sizeof head);
    }
    _exit(0);
  }
  close(fds[1]);
  int32_t head[3];
  int rc = REF_CRASHED;
  if (read_all(fds[0], head, sizeof head)) {
    rc = head[0];
    if (rc == REF_OK) {
      *w = (uint32_t)head[1];
      *h = (uint32_t)head[2];
      size_t n = (size_t)head[1] * (size_t)head[2];
      if (n > cap_pixels) {
        rc = REF_NO_ROOM;
      } else if (!read_all(fds[0], argb, 4 * n)) {
        rc = REF_CRASHED;
      }
    }
  }
  close(fds[0]);
  int st = 0;
  waitpid(pid, &st, 0);
  // a child that produced its pixels and then died while unwinding (heap damage from the
  // unchecked palette rows) still counts as crashed
  if (rc == REF_OK && !(WIFEXITED(st) && WEXITSTATUS(st) == 0)) rc = REF_CRASHED;
  return rc;
}

}  // extern "C"
