// kt_host_tool — tiny CLI over the host-side parsers, used by the CPU tests to cross-check them against
// the independent Python implementations (kube_throttler_amd/quantity.py).
//   kt_host_tool quantity <text>...   -> "<nano value> <DecimalSI text> <canonical text in its own format>" or "error: ..."
//   kt_host_tool time <rfc3339>...    -> "<sec> <nsec>" or "error: ..."
//   kt_host_tool label <text>...      -> "<valid as label key 0/1> <valid as label value 0/1>"
#include <cstdio>
#include <cstring>
#include <string>

#include "kt_host.hpp"

static std::string i128(__int128 v) {
  if (v == 0) return "0";
  bool neg = v < 0;
  if (neg) v = -v;
  std::string s;
  while (v > 0) s.insert(s.begin(), (char)('0' + (int)(v % 10))), v /= 10;
  return neg ? "-" + s : s;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  for (int i = 2; i < argc; ++i) {
    std::string err;
    if (!strcmp(argv[1], "quantity")) {
      kth::Quantity q;
      if (kth::ParseQuantity(argv[i], &q, &err)) printf("%s %s %s\n", i128(q.nano).c_str(), kth::FormatDecimalSI(q).c_str(), kth::FormatQuantity(q).c_str());
      else printf("error: %s\n", err.c_str());
    } else if (!strcmp(argv[1], "label")) {
      printf("%d %d\n", kth::ValidLabelKey(argv[i]) ? 1 : 0, kth::ValidLabelValue(argv[i]) ? 1 : 0);
    } else {
      int64_t s;
      int32_t ns;
      if (kth::ParseRFC3339(argv[i], &s, &ns, &err)) printf("%lld %d\n", (long long)s, ns);
      else printf("error: %s\n", err.c_str());
    }
  }
  return 0;
}
