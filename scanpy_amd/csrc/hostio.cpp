// Host-side byte codecs of the on-disk readers (scanpy_amd/_hdf5.py): no device work, callable without a GPU.
// ctypes releases the GIL around these calls, so the reader's thread pool decodes chunks in parallel.
#include "common.h"

#include <cstring>

// LZF (Marc Lehmann's liblzf format, the codec behind h5py's `compression='lzf'` filter 32000): a stream of
//   ctrl < 32            literal run of ctrl + 1 bytes
//   ctrl >= 32           back reference: len = ctrl >> 5 (7 = extended by the next byte), then + 2;
//                        distance = ((ctrl & 31) << 8 | next byte) + 1 bytes behind the write position
extern "C" int64_t scamd_lzf_decompress(const void* src_, size_t src_len, void* dst_, size_t dst_cap) {
  if ((!src_ && src_len) || (!dst_ && dst_cap)) {
    scamd::set_error("scamd_lzf_decompress: null buffer");
    return SCAMD_EINVAL;
  }
  const uint8_t* ip = static_cast<const uint8_t*>(src_);
  const uint8_t* const ie = ip + src_len;
  uint8_t* const out = static_cast<uint8_t*>(dst_);
  uint8_t* op = out;
  uint8_t* const oe = out + dst_cap;
  while (ip < ie) {
    unsigned ctrl = *ip++;
    if (ctrl < 32) {
      size_t run = ctrl + 1;
      if (run > static_cast<size_t>(ie - ip) || run > static_cast<size_t>(oe - op)) {
        scamd::set_error("scamd_lzf_decompress: literal run leaves the buffers");
        return SCAMD_EINVAL;
      }
      std::memcpy(op, ip, run);
      ip += run;
      op += run;
    } else {
      size_t len = ctrl >> 5;
      if (len == 7) {
        if (ip >= ie) {
          scamd::set_error("scamd_lzf_decompress: truncated stream");
          return SCAMD_EINVAL;
        }
        len += *ip++;
      }
      if (ip >= ie) {
        scamd::set_error("scamd_lzf_decompress: truncated stream");
        return SCAMD_EINVAL;
      }
      size_t dist = (static_cast<size_t>(ctrl & 31) << 8 | *ip++) + 1;
      len += 2;
      if (dist > static_cast<size_t>(op - out) || len > static_cast<size_t>(oe - op)) {
        scamd::set_error("scamd_lzf_decompress: back reference leaves the buffers");
        return SCAMD_EINVAL;
      }
      const uint8_t* ref = op - dist;
      for (size_t i = 0; i < len; ++i) op[i] = ref[i];  // byte-wise: the ranges may overlap (run-length encoding)
      op += len;
    }
  }
  return static_cast<int64_t>(op - out);
}

// Inverse of the HDF5 shuffle filter: src holds elem_size byte planes of n_elem bytes each; dst gets the elements.
extern "C" int scamd_unshuffle(const void* src_, void* dst_, size_t n_elem, int elem_size) {
  if (elem_size < 1 || ((!src_ || !dst_) && n_elem)) {
    scamd::set_error("scamd_unshuffle: bad arguments");
    return SCAMD_EINVAL;
  }
  const uint8_t* src = static_cast<const uint8_t*>(src_);
  uint8_t* dst = static_cast<uint8_t*>(dst_);
  const size_t es = static_cast<size_t>(elem_size);
  if (es == 4) {
    const uint8_t *p0 = src, *p1 = src + n_elem, *p2 = src + 2 * n_elem, *p3 = src + 3 * n_elem;
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    if ((reinterpret_cast<uintptr_t>(dst) & 3) == 0) {
      for (size_t i = 0; i < n_elem; ++i)
        d[i] = static_cast<uint32_t>(p0[i]) | static_cast<uint32_t>(p1[i]) << 8 | static_cast<uint32_t>(p2[i]) << 16 |
               static_cast<uint32_t>(p3[i]) << 24;
      return 0;
    }
  }
  constexpr size_t kBlock = 4096;  // keep one block of every plane and of the output in L1/L2
  for (size_t b = 0; b < n_elem; b += kBlock) {
    const size_t m = n_elem - b < kBlock ? n_elem - b : kBlock;
    for (size_t j = 0; j < es; ++j) {
      const uint8_t* plane = src + j * n_elem + b;
      uint8_t* o = dst + b * es + j;
      for (size_t i = 0; i < m; ++i) o[i * es] = plane[i];
    }
  }
  return 0;
}
