// vp8l_split.h -- how a batch of independent images is dealt out to several GPUs (host code).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace vp8l {

// Contiguous blocks of items, one per part, balanced by weight (pixels): part p gets the items
// [first[p], first[p + 1]).  An item goes to the part in which its middle falls when the total weight is cut
// into equal shares, so every part's weight is within one item of its share; parts may be empty when there
// are fewer items than parts.  Images are independent, so nothing is exchanged between the parts.
inline std::vector<size_t> split_by_weight(const std::vector<uint64_t> &weight, size_t parts) {
  std::vector<size_t> first(parts + 1, weight.size());
  first[0] = 0;
  if (parts == 0) return first;
  long double total = 0;
  for (uint64_t w : weight) total += (long double)(w ? w : 1);
  long double acc = 0;
  size_t part = 0;
  for (size_t i = 0; i < weight.size(); i++) {
    const long double w = (long double)(weight[i] ? weight[i] : 1);
    const long double mid = acc + w / 2;
    size_t want = (size_t)(mid * (long double)parts / total);
    if (want >= parts) want = parts - 1;
    while (part < want) first[++part] = i;
    acc += w;
  }
  while (part < parts) first[++part] = weight.size();
  return first;
}

}  // namespace vp8l
