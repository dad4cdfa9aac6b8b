#include <mutex>
#include <set>
// kt_host.cpp — see kt_host.hpp.  Host-side mirror of pkg/scheduler_plugin/plugin.go over the C-ABI.
#include "kt_host.hpp"

#include <algorithm>
#include <cctype>
#include <cstring>

namespace kth {

const char* const PluginName = "kube-throttler";

// =====================================================================================================
// resource.Quantity
// =====================================================================================================
namespace {
__int128 pow10_128(int e) {
  __int128 r = 1;
  while (e-- > 0) r *= 10;
  return r;
}
}  // namespace

bool ParseQuantity(const std::string& text, Quantity* out, std::string* err) {
  // <quantity> ::= [+-]? digits[.digits]? | [+-]? .digits   followed by  Ki|Mi|Gi|Ti|Pi|Ei | n|u|m|""|k|M|G|T|P|E | e[+-]?N
  size_t i = 0;
  const size_t n = text.size();
  bool neg = false;
  if (i < n && (text[i] == '+' || text[i] == '-')) neg = text[i++] == '-';
  __int128 mant = 0;
  int frac_digits = 0, digits = 0;
  while (i < n && std::isdigit((unsigned char)text[i])) mant = mant * 10 + (text[i++] - '0'), ++digits;
  if (i < n && text[i] == '.') {
    ++i;
    while (i < n && std::isdigit((unsigned char)text[i])) mant = mant * 10 + (text[i++] - '0'), ++digits, ++frac_digits;
  }
  if (digits == 0 || digits > 36) {
    if (err) *err = "quantities must match the regular expression '^([+-]?[0-9.]+)([eEinumkKMGTP]*[-+]?[0-9]*)$'";
    return false;
  }
  const std::string suf = text.substr(i);
  int bin = -1, dec = 0;
  static const char* kBin[] = {"Ki", "Mi", "Gi", "Ti", "Pi", "Ei"};
  for (int k = 0; k < 6; ++k)
    if (suf == kBin[k]) bin = 10 * (k + 1);
  if (bin < 0) {
    if (suf == "n") dec = -9;
    else if (suf == "u") dec = -6;
    else if (suf == "m") dec = -3;
    else if (suf.empty()) dec = 0;
    else if (suf == "k") dec = 3;
    else if (suf == "M") dec = 6;
    else if (suf == "G") dec = 9;
    else if (suf == "T") dec = 12;
    else if (suf == "P") dec = 15;
    else if (suf == "E") dec = 18;
    else if ((suf[0] == 'e' || suf[0] == 'E') && suf.size() > 1) {
      size_t j = 1;
      bool eneg = false;
      if (suf[j] == '+' || suf[j] == '-') eneg = suf[j++] == '-';
      if (j >= suf.size()) { if (err) *err = "unable to parse quantity's suffix"; return false; }
      int ev = 0;
      for (; j < suf.size(); ++j) {
        if (!std::isdigit((unsigned char)suf[j]) || ev > 100) { if (err) *err = "unable to parse quantity's suffix"; return false; }
        ev = ev * 10 + (suf[j] - '0');
      }
      dec = eneg ? -ev : ev;
    } else {
      if (err) *err = "unable to parse quantity's suffix";
      return false;
    }
  }
  // value = mant * 10^-frac * (2^bin | 10^dec); in nano units: * 10^9
  __int128 num = mant;
  int exp10 = 9 - frac_digits + (bin < 0 ? dec : 0);
  if (bin >= 0) num <<= bin;
  __int128 nano;
  if (exp10 >= 0) {
    if (exp10 > 30) { if (err) *err = "quantity out of range"; return false; }
    nano = num * pow10_128(exp10);
  } else {
    if (-exp10 > 36) { nano = num != 0 ? 1 : 0; }
    else {
      const __int128 d = pow10_128(-exp10);
      nano = num / d;
      if (num % d != 0) nano += 1;  // round away from zero (magnitude; sign applied below)
    }
  }
  out->nano = neg ? -nano : nano;
  out->format = bin >= 0 ? Format::BinarySI
                         : (!suf.empty() && (suf[0] == 'e' || (suf[0] == 'E' && suf.size() > 1))) ? Format::DecimalExponent
                                                                                                   : Format::DecimalSI;
  return true;
}

bool ScaledValue(const Quantity& q, int scale, int64_t* out) {
  // value / 10^scale = nano * 10^(-9 - scale)
  const int e = -9 - scale;
  __int128 v = q.nano;
  if (e >= 0) {
    if (e > 18) return false;
    v *= pow10_128(e);
  } else {
    const __int128 d = pow10_128(-e);
    if (v % d != 0) return false;
    v /= d;
  }
  if (v > (__int128)INT64_MAX || v < (__int128)INT64_MIN) return false;
  *out = (int64_t)v;
  return true;
}

std::string FormatDecimalSI(const Quantity& q) {
  if (q.nano == 0) return "0";
  static const struct { int e; const char* s; } kSuf[] = {{27, "E"}, {24, "P"}, {21, "T"}, {18, "G"}, {15, "M"}, {12, "k"},
                                                          {9, ""},   {6, "m"},  {3, "u"},  {0, "n"}};
  for (auto& sf : kSuf) {
    const __int128 d = pow10_128(sf.e);
    if (q.nano % d == 0) {
      __int128 v = q.nano / d;
      const bool neg = v < 0;
      if (neg) v = -v;
      std::string digits;
      do { digits.insert(digits.begin(), (char)('0' + (int)(v % 10))); v /= 10; } while (v > 0);
      return (neg ? "-" : "") + digits + sf.s;
    }
  }
  return "?";
}

namespace {
std::string digits_of(__int128 v) {
  const bool neg = v < 0;
  if (neg) v = -v;
  std::string digits;
  do { digits.insert(digits.begin(), (char)('0' + (int)(v % 10))); v /= 10; } while (v > 0);
  return (neg ? "-" : "") + digits;
}
}  // namespace

std::string FormatQuantity(const Quantity& q) {
  if (q.nano == 0) return "0";
  Format f = q.format;
  const __int128 one = pow10_128(9);
  if (f == Format::BinarySI) {
    const __int128 mag = q.nano < 0 ? -q.nano : q.nano;
    if (mag < 1024 * one || q.nano % one != 0) f = Format::DecimalSI;  // small or fractional: shown as DecimalSI
  }
  if (f == Format::BinarySI) {
    static const char* kBin[] = {"", "Ki", "Mi", "Gi", "Ti", "Pi", "Ei"};
    __int128 v = q.nano / one;
    int e = 0;
    while (e < 6 && v % 1024 == 0) v /= 1024, ++e;
    return digits_of(v) + kBin[e];
  }
  if (f == Format::DecimalSI) return FormatDecimalSI(q);
  // DecimalExponent: mantissa without factors of 10, then the exponent lowered to a multiple of 3
  __int128 v = q.nano;
  int e = -9;
  while (v % 10 == 0) v /= 10, ++e;
  int r = ((e % 3) + 3) % 3;
  for (; r > 0; --r) v *= 10, --e;
  return e == 0 ? digits_of(v) : digits_of(v) + "e" + std::to_string(e);
}

std::map<std::string, std::string> ThrottleStatus::UsedStrings() const {
  std::map<std::string, std::string> out;
  for (auto& kv : used) out[kv.first] = FormatQuantity(kv.second);
  return out;
}

bool StatusSemanticEqual(const ThrottleStatus& a, const ThrottleStatus& b) {
  if (a.usedHasCounts != b.usedHasCounts || (a.usedHasCounts && a.usedPod != b.usedPod)) return false;
  if (a.throttledPod != b.throttledPod || a.throttledRequests != b.throttledRequests) return false;
  if (a.used.size() != b.used.size()) return false;
  for (auto& kv : a.used) {
    auto it = b.used.find(kv.first);
    if (it == b.used.end() || it->second.nano != kv.second.nano) return false;  // Format is not part of Cmp
  }
  return true;
}

// =====================================================================================================
// RFC3339
// =====================================================================================================
namespace {
int64_t days_from_civil(int64_t y, unsigned m, unsigned d) {  // proleptic Gregorian, days since 1970-01-01
  y -= m <= 2;
  const int64_t era = (y >= 0 ? y : y - 399) / 400;
  const unsigned yoe = (unsigned)(y - era * 400);
  const unsigned doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
  const unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  return era * 146097 + (int64_t)doe - 719468;
}
bool num(const std::string& s, size_t pos, size_t len, int* out) {
  if (pos + len > s.size()) return false;
  int v = 0;
  for (size_t i = 0; i < len; ++i) {
    if (!std::isdigit((unsigned char)s[pos + i])) return false;
    v = v * 10 + (s[pos + i] - '0');
  }
  *out = v;
  return true;
}
}  // namespace

bool ParseRFC3339(const std::string& t, int64_t* sec, int32_t* nsec, std::string* err) {
  auto fail = [&]() {
    if (err) *err = "parsing time \"" + t + "\" as \"2006-01-02T15:04:05Z07:00\": cannot parse";
    return false;
  };
  int Y, M, D, h, m, s;
  if (!num(t, 0, 4, &Y) || t.size() < 20 || t[4] != '-' || !num(t, 5, 2, &M) || t[7] != '-' || !num(t, 8, 2, &D) ||
      (t[10] != 'T' && t[10] != 't') || !num(t, 11, 2, &h) || t[13] != ':' || !num(t, 14, 2, &m) || t[16] != ':' ||
      !num(t, 17, 2, &s))
    return fail();
  static const int kDays[] = {31, 29, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
  if (M < 1 || M > 12 || D < 1 || D > kDays[M - 1] || h > 23 || m > 59 || s > 59) return fail();
  if (M == 2 && D == 29 && !((Y % 4 == 0 && Y % 100 != 0) || Y % 400 == 0)) return fail();
  size_t i = 19;
  int64_t ns = 0;
  if (t[i] == '.' || t[i] == ',') {
    ++i;
    int nd = 0;
    while (i < t.size() && std::isdigit((unsigned char)t[i])) {
      if (nd < 9) ns = ns * 10 + (t[i] - '0'), ++nd;
      ++i;
    }
    if (nd == 0) return fail();
    while (nd++ < 9) ns *= 10;
  }
  int64_t off = 0;
  if (i < t.size() && (t[i] == 'Z' || t[i] == 'z') && i + 1 == t.size()) {
  } else if (i + 6 == t.size() && (t[i] == '+' || t[i] == '-') && t[i + 3] == ':') {
    int oh, om;
    if (!num(t, i + 1, 2, &oh) || !num(t, i + 4, 2, &om) || oh > 23 || om > 59) return fail();
    off = (oh * 3600 + om * 60) * (t[i] == '+' ? 1 : -1);
  } else {
    return fail();
  }
  *sec = days_from_civil(Y, (unsigned)M, (unsigned)D) * 86400 + h * 3600 + m * 60 + s - off;
  *nsec = (int32_t)ns;
  return true;
}

// =====================================================================================================
// plugin
// =====================================================================================================
namespace {

bool valid_name_part(const std::string& s) {  // qualified-name "name" part: [A-Za-z0-9]([-A-Za-z0-9_.]*[A-Za-z0-9])?, <= 63
  if (s.empty() || s.size() > 63) return false;
  if (!std::isalnum((unsigned char)s.front()) || !std::isalnum((unsigned char)s.back())) return false;
  for (char c : s)
    if (!std::isalnum((unsigned char)c) && c != '-' && c != '_' && c != '.') return false;
  return true;
}
bool valid_label_key(const std::string& k) {
  const size_t slash = k.find('/');
  if (slash == std::string::npos) return valid_name_part(k);
  if (k.find('/', slash + 1) != std::string::npos) return false;
  const std::string prefix = k.substr(0, slash);
  if (prefix.empty() || prefix.size() > 253) return false;
  // DNS-1123 subdomain
  bool start = true;
  for (size_t i = 0; i < prefix.size(); ++i) {
    const char c = prefix[i];
    if (c == '.') {
      if (start || prefix[i - 1] == '-') return false;
      start = true;
      continue;
    }
    if (!(std::islower((unsigned char)c) || std::isdigit((unsigned char)c) || c == '-')) return false;
    if (start && c == '-') return false;
    start = false;
  }
  if (start || prefix.back() == '-') return false;
  return valid_name_part(k.substr(slash + 1));
}
bool valid_label_value(const std::string& v) { return v.empty() || valid_name_part(v); }
}  // namespace
// exported for the CPU cross-check against the Python restatement (objects.py)
bool ValidLabelKey(const std::string& k) { return valid_label_key(k); }
bool ValidLabelValue(const std::string& v) { return valid_label_value(v); }
namespace {

template <class K>
struct RowTable {  // key -> dense row with a free list
  std::unordered_map<K, int64_t> row_of;
  std::vector<int64_t> free_rows;
  int64_t next = 0, cap = 0;
  int64_t find(const K& k) const {
    auto it = row_of.find(k);
    return it == row_of.end() ? -1 : it->second;
  }
  int64_t acquire(const K& k) {
    int64_t r = find(k);
    if (r >= 0) return r;
    if (!free_rows.empty()) {
      r = free_rows.back();
      free_rows.pop_back();
    } else if (next < cap) {
      r = next++;
    } else {
      return -1;
    }
    row_of[k] = r;
    return r;
  }
  void release(const K& k) {
    auto it = row_of.find(k);
    if (it == row_of.end()) return;
    free_rows.push_back(it->second);
    row_of.erase(it);
  }
};

struct DenseAmount {
  int64_t v[KT_MAX_DIMS] = {0};
  uint32_t present = 0;
  int64_t count = 0;
  uint8_t has_count = 0;
};

}  // namespace

struct KubeThrottler::Impl {
  // PreFilter / Reserve run on the scheduling goroutine, Unreserve on binding goroutines, the event handlers on informer
  // goroutines (plugin.go:148-257): every public method takes this lock (recursive: the methods call each other)
  std::recursive_mutex mu;
  PluginArgs args;
  kt_engine* e = nullptr;
  int D = KT_MAX_DIMS;
  std::map<std::string, std::pair<int, int>> dims;  // resource name -> (dimension, scale)
  // resource name -> Format of the first non-zero quantity seen under that name: what a `used` sum inherits through
  // Quantity.Add when every pod writes the resource in one suffix family (the reference's result otherwise depends on
  // the lister's pod order)
  std::map<std::string, Format> dim_format;
  std::map<std::string, ThrottleStatus> written;  // thr_key -> status last handed to UpdateStatus
  std::unordered_map<std::string, uint32_t> key_ids;
  std::unordered_map<std::string, uint32_t> pair_ids;
  RowTable<std::string> ns_rows, pod_rows, thr_rows;
  std::vector<Throttle> thr_by_row;
  std::vector<uint8_t> thr_live;
  std::vector<std::vector<std::string>> thr_msgs;
  std::unordered_map<std::string, Pod> pods;
  // reserved cache: throttle row -> pod key -> amount of the pod (reserved_resource_amounts.go:32-41)
  std::map<int32_t, std::map<std::string, DenseAmount>> reserved;
  std::vector<uint8_t> last_status;
  int64_t last_row_pending = -1;  // pod row whose status row LastStatusOf still has to fetch (PreFilter allowed it on the summary)

  uint32_t key_id(const std::string& k) { return key_ids.emplace(k, (uint32_t)key_ids.size() + 1).first->second; }
  uint32_t pair_id(const std::string& k, const std::string& v) {
    std::string kv = k;
    kv.push_back('\0');
    kv += v;
    return pair_ids.emplace(kv, (uint32_t)pair_ids.size() + 1).first->second;
  }
  static std::string thr_key(const Throttle& t) { return (t.cluster ? "C:" : "T:") + t.Key(); }

  bool dim_of(const std::string& name, int* dim, int* scale, std::string* err) {
    auto it = dims.find(name);
    if (it == dims.end()) {
      if ((int)dims.size() >= D) {
        if (err) *err = "more than " + std::to_string(D) + " distinct resource names";
        return false;
      }
      int sc = name == "cpu" ? -3 : 0;
      auto s = args.resourceScales.find(name);
      if (s != args.resourceScales.end()) sc = s->second;
      it = dims.emplace(name, std::make_pair((int)dims.size(), sc)).first;
    }
    *dim = it->second.first;
    *scale = it->second.second;
    return true;
  }
  bool fill_row(const ResourceList& rl, int64_t* v, uint32_t* present, std::string* err) {
    for (auto& kv : rl) {
      Quantity q;
      int dim, scale;
      if (!ParseQuantity(kv.second, &q, err) || !dim_of(kv.first, &dim, &scale, err)) return false;
      int64_t x;
      if (!ScaledValue(q, scale, &x)) {
        if (err) *err = "quantity " + kv.second + " of " + kv.first + " is not representable at scale 1e" + std::to_string(scale);
        return false;
      }
      v[dim] = x;
      *present |= 1u << dim;
      if (q.nano != 0) dim_format.emplace(kv.first, q.format);
    }
    return true;
  }
  bool fill_amount(const ResourceAmount& a, DenseAmount* d, std::string* err) {
    *d = DenseAmount();
    d->has_count = a.hasCounts;
    d->count = a.hasCounts ? a.pod : 0;
    return fill_row(a.requests, d->v, &d->present, err);
  }
  std::string engine_error(int32_t rc) { return "kt: " + std::to_string(rc) + ": " + kt_last_error(e); }

  bool push_reserved(int32_t row, std::string* err) {
    DenseAmount tot;
    auto it = reserved.find(row);
    if (it != reserved.end())
      for (auto& kv : it->second) {  // podResourceAmountMap.totalResoruceAmount (reserved_resource_amounts.go:148-156)
        tot.has_count = 1;
        tot.count += 1;
        tot.present |= kv.second.present;
        for (int d = 0; d < D; ++d) tot.v[d] += kv.second.v[d];
      }
    kt_amounts am{tot.v, &tot.present, &tot.count, &tot.has_count};
    int32_t rc = kt_set_reserved(e, 1, &row, &am);
    if (rc != KT_OK) {
      if (err) *err = engine_error(rc);
      return false;
    }
    return true;
  }
};

KubeThrottler::~KubeThrottler() {
  if (p_ && p_->e) kt_engine_destroy(p_->e);
}

std::unique_ptr<KubeThrottler> NewPlugin(const PluginArgs& args, std::string* err) {
  // DecodePluginArgs (plugin_args.go:46-51)
  if (args.name.empty()) {
    if (err) *err = "Name must not be empty";
    return nullptr;
  }
  if (args.targetSchedulerName.empty()) {
    if (err) *err = "TargetSchedulerName must not be empty";
    return nullptr;
  }
  std::unique_ptr<KubeThrottler> k(new KubeThrottler());
  k->p_.reset(new KubeThrottler::Impl());
  auto& p = *k->p_;
  p.args = args;
  kt_config cfg{};
  cfg.n_dims = p.D;
  cfg.max_labels = args.maxLabels;
  cfg.pod_capacity = args.podCapacity;
  cfg.throttle_capacity = args.throttleCapacity;
  cfg.namespace_capacity = args.namespaceCapacity;
  cfg.device = -1;
  cfg.kernel_variant = 0;
  int32_t rc = kt_engine_create(&cfg, &p.e);
  if (rc != KT_OK) {
    if (err) *err = std::string("kt_engine_create: ") + std::to_string(rc) + ": " + kt_last_error(nullptr);
    return nullptr;
  }
  p.ns_rows.cap = args.namespaceCapacity;
  p.pod_rows.cap = args.podCapacity;
  p.thr_rows.cap = args.throttleCapacity;
  p.thr_by_row.resize((size_t)args.throttleCapacity);
  p.thr_live.assign((size_t)args.throttleCapacity, 0);
  p.thr_msgs.resize((size_t)args.throttleCapacity);
  for (auto& kv : args.resourceScales) {
    int d, s;
    p.dim_of(kv.first, &d, &s, nullptr);
  }
  return k;
}

// ---------------------------------------------------------------------------------------------------
// informer feed
// ---------------------------------------------------------------------------------------------------
bool KubeThrottler::OnNamespaceAdd(const Namespace& ns, std::string* err) {
  std::lock_guard<std::recursive_mutex> lk(p_->mu);
  auto& p = *p_;
  const int64_t row = p.ns_rows.acquire(ns.name);
  if (row < 0) { if (err) *err = "namespace capacity exhausted"; return false; }
  std::vector<uint32_t> keys, pairs;
  for (auto& kv : ns.labels) keys.push_back(p.key_id(kv.first)), pairs.push_back(p.pair_id(kv.first, kv.second));
  uint32_t off[2] = {0, (uint32_t)keys.size()};
  uint8_t valid = 1;
  keys.push_back(0), pairs.push_back(0);
  kt_snapshot b{};
  b.D = p.D;
  b.n_ns = 1;
  b.ns_valid = &valid;
  b.ns_label_off = off;
  b.ns_label_key = keys.data();
  b.ns_label_pair = pairs.data();
  const int32_t r32 = (int32_t)row;
  int32_t rc = kt_upsert_namespaces(p.e, &b, &r32);
  if (rc != KT_OK) { if (err) *err = p.engine_error(rc); return false; }
  return true;
}

bool KubeThrottler::OnNamespaceDelete(const std::string& name, std::string* err) {
  std::lock_guard<std::recursive_mutex> lk(p_->mu);
  auto& p = *p_;
  const int64_t row = p.ns_rows.find(name);
  if (row < 0) return true;
  // keep the id (pods may still reference it): the Namespace OBJECT is gone => mark invalid
  uint32_t off[2] = {0, 0}, zero = 0;
  uint8_t valid = 0;
  kt_snapshot b{};
  b.D = p.D;
  b.n_ns = 1;
  b.ns_valid = &valid;
  b.ns_label_off = off;
  b.ns_label_key = &zero;
  b.ns_label_pair = &zero;
  const int32_t r32 = (int32_t)row;
  int32_t rc = kt_upsert_namespaces(p.e, &b, &r32);
  if (rc != KT_OK) { if (err) *err = p.engine_error(rc); return false; }
  return true;
}

bool KubeThrottler::OnPodAdd(const Pod& pod, std::string* err) {
  std::lock_guard<std::recursive_mutex> lk(p_->mu);
  auto& p = *p_;
  // the pod's namespace id must exist even when no Namespace object was seen (then it stays invalid)
  int64_t ns_row = p.ns_rows.find(pod.ns);
  if (ns_row < 0) {
    ns_row = p.ns_rows.acquire(pod.ns);
    if (ns_row < 0) { if (err) *err = "namespace capacity exhausted"; return false; }
    if (!OnNamespaceDelete(pod.ns, err)) return false;  // registers the id as "no Namespace object"
  }
  if ((int)pod.labels.size() > p.args.maxLabels) { if (err) *err = "pod has more labels than maxLabels"; return false; }
  const int64_t row = p.pod_rows.acquire(pod.Key());
  if (row < 0) { if (err) *err = "pod capacity exhausted"; return false; }
  const int D = p.D;
  std::vector<uint32_t> keys, pairs;
  for (auto& kv : pod.labels) keys.push_back(p.key_id(kv.first)), pairs.push_back(p.pair_id(kv.first, kv.second));
  uint32_t loff[2] = {0, (uint32_t)keys.size()};
  keys.push_back(0), pairs.push_back(0);
  const size_t nc = pod.containers.size() + pod.initContainers.size();
  std::vector<uint8_t> c_init(nc + 1, 0);
  std::vector<uint32_t> c_present(nc + 1, 0);
  std::vector<int64_t> c_req((nc + 1) * D, 0);
  size_t c = 0;
  for (auto& ic : pod.initContainers) {
    c_init[c] = 1;
    if (!p.fill_row(ic.requests, &c_req[c * D], &c_present[c], err)) return false;
    ++c;
  }
  for (auto& ct : pod.containers) {
    if (!p.fill_row(ct.requests, &c_req[c * D], &c_present[c], err)) return false;
    ++c;
  }
  uint32_t coff[2] = {0, (uint32_t)nc};
  std::vector<int64_t> ovh(D, 0);
  uint32_t ovh_present = 0;
  if (pod.hasOverhead) {
    if (!p.fill_row(pod.overhead, ovh.data(), &ovh_present, err)) return false;
    ovh_present |= 0x80000000u;
  }
  uint32_t ns32 = (uint32_t)ns_row;
  uint32_t flags = KT_POD_VALID;
  if (pod.schedulerName == p.args.targetSchedulerName) flags |= KT_POD_SCHED_MATCH;
  if (!pod.nodeName.empty()) flags |= KT_POD_SCHEDULED;
  if (pod.phase == "Succeeded" || pod.phase == "Failed") flags |= KT_POD_FINISHED;
  kt_snapshot b{};
  b.D = D;
  b.L = p.args.maxLabels;
  b.n_pods = 1;
  b.pod_ns = &ns32;
  b.pod_flags = &flags;
  b.pod_label_off = loff;
  b.pod_label_key = keys.data();
  b.pod_label_pair = pairs.data();
  b.pod_ctr_off = coff;
  b.ctr_init = c_init.data();
  b.ctr_present = c_present.data();
  b.ctr_req = c_req.data();
  b.pod_ovh_present = &ovh_present;
  b.pod_ovh = ovh.data();
  int32_t rc = kt_upsert_pods(p.e, &b, &row);
  if (rc != KT_OK) { if (err) *err = p.engine_error(rc); return false; }
  p.pods[pod.Key()] = pod;
  return true;
}

bool KubeThrottler::OnPodDelete(const std::string& key, std::string* err) {
  std::lock_guard<std::recursive_mutex> lk(p_->mu);
  auto& p = *p_;
  const int64_t row = p.pod_rows.find(key);
  if (row < 0) return true;
  auto known = p.pods.find(key);
  if (known != p.pods.end() && known->second.schedulerName == p.args.targetSchedulerName && !known->second.nodeName.empty())
    Unreserve(known->second);  // "observe the deleted pod is now scheduled. controller should unreserve it."
  int32_t rc = kt_delete_pods(p.e, 1, &row);
  if (rc != KT_OK) { if (err) *err = p.engine_error(rc); return false; }
  p.pod_rows.release(key);
  p.pods.erase(key);
  return true;
}

namespace {
struct ReqPool {
  std::vector<uint8_t> op;
  std::vector<uint32_t> key, val_off{0}, val;
  void add(uint8_t o, uint32_t k, const std::vector<uint32_t>& vals) {
    op.push_back(o);
    key.push_back(k);
    val.insert(val.end(), vals.begin(), vals.end());
    val_off.push_back((uint32_t)val.size());
  }
  kt_reqs view() {
    if (op.empty()) op.push_back(0), key.push_back(0);
    if (val.empty()) val.push_back(0);
    kt_reqs r;
    r.n = (uint32_t)(val_off.size() - 1);
    r.op = op.data();
    r.key = key.data();
    r.val_off = val_off.data();
    r.val = val.data();
    return r;
  }
};
}  // namespace

bool KubeThrottler::OnThrottleAdd(const Throttle& thr, std::string* err) {
  std::lock_guard<std::recursive_mutex> lk(p_->mu);
  auto& p = *p_;
  const int D = p.D;
  int64_t ns_row = 0;
  if (!thr.cluster) {
    ns_row = p.ns_rows.find(thr.ns);
    if (ns_row < 0) {
      ns_row = p.ns_rows.acquire(thr.ns);
      if (ns_row < 0) { if (err) *err = "namespace capacity exhausted"; return false; }
      if (!OnNamespaceDelete(thr.ns, err)) return false;
    }
  }
  const std::string tk = Impl::thr_key(thr);
  const bool existed = p.thr_rows.find(tk) >= 0;
  const int64_t row = p.thr_rows.acquire(tk);
  if (row < 0) { if (err) *err = "throttle capacity exhausted"; return false; }
  DenseAmount spec;
  if (!p.fill_amount(thr.threshold, &spec, err)) return false;
  // overrides: parse instants once; unparsable ones are flagged and produce the reference's messages
  const size_t no = thr.overrides.size();
  std::vector<int64_t> ob(no + 1, KT_ZERO_TIME_S), oe(no + 1, KT_ZERO_TIME_S), ov((no + 1) * D, 0), ocount(no + 1, 0);
  std::vector<int32_t> obn(no + 1, 0), oen(no + 1, 0);
  std::vector<uint8_t> oflags(no + 1, 0), ohas(no + 1, 0);
  std::vector<uint32_t> opresent(no + 1, 0);
  std::vector<std::string> msgs;
  for (size_t j = 0; j < no; ++j) {
    const auto& o = thr.overrides[j];
    std::string perr;
    bool bad = false, begin_ok = false;
    if (!o.begin.empty() && !ParseRFC3339(o.begin, &ob[j], &obn[j], &perr)) {
      bad = true;
      msgs.push_back("index " + std::to_string(j) + ": Failed to parse Begin: " + perr);
    } else if (!o.end.empty() && !ParseRFC3339(o.end, &oe[j], &oen[j], &perr)) {
      bad = begin_ok = true;
      msgs.push_back("index " + std::to_string(j) + ": Failed to parse End: " + perr);
    }
    if (bad) {
      oflags[j] |= KT_OVR_PARSE_ERROR;
      oe[j] = KT_ZERO_TIME_S, oen[j] = 0;
      if (begin_ok) oflags[j] |= KT_OVR_BEGIN_PARSED;  // NextOverrideHappensIn still counts the begin instant
      else ob[j] = KT_ZERO_TIME_S, obn[j] = 0;
    }
    DenseAmount a;
    if (!p.fill_amount(o.threshold, &a, err)) return false;
    std::memcpy(&ov[j * D], a.v, sizeof(int64_t) * D);
    opresent[j] = a.present;
    ocount[j] = a.count;
    ohas[j] = a.has_count;
  }
  uint64_t spec_fp = 0;
  for (auto& m : msgs)
    for (char ch : m) spec_fp = (spec_fp ^ (uint8_t)ch) * 1099511628211ull + 0x9E3779B97F4A7C15ull;
  if (!msgs.empty() && spec_fp == 0) spec_fp = 1;
  // selector terms -> requirement pools; LabelSelectorAsSelector validation decides the INVALID flags
  ReqPool preq, nreq;
  const size_t nt = thr.selectorTerms.size();
  std::vector<uint8_t> tflags(nt + 1, 0);
  std::vector<uint32_t> tpo(nt + 1, 0), tno(nt + 1, 0);
  auto convert = [&](const LabelSelector& sel, ReqPool& pool) -> bool {  // returns validity
    bool ok = true;
    for (auto& kv : sel.matchLabels) {
      if (!valid_label_key(kv.first) || !valid_label_value(kv.second)) ok = false;
      pool.add(KT_OP_IN, p.key_id(kv.first), {p.pair_id(kv.first, kv.second)});
    }
    for (auto& e : sel.matchExpressions) {
      int op = e.op == "In" ? KT_OP_IN : e.op == "NotIn" ? KT_OP_NOT_IN : e.op == "Exists" ? KT_OP_EXISTS
               : e.op == "DoesNotExist" ? KT_OP_DOES_NOT_EXIST : -1;
      if (op < 0) { ok = false; continue; }
      if ((op == KT_OP_IN || op == KT_OP_NOT_IN) && e.values.empty()) ok = false;
      if ((op == KT_OP_EXISTS || op == KT_OP_DOES_NOT_EXIST) && !e.values.empty()) ok = false;
      if (!valid_label_key(e.key)) ok = false;
      std::vector<uint32_t> vals;
      for (auto& v : e.values) {
        if (!valid_label_value(v)) ok = false;
        vals.push_back(p.pair_id(e.key, v));
      }
      pool.add((uint8_t)op, p.key_id(e.key), vals);
    }
    return ok;
  };
  for (size_t j = 0; j < nt; ++j) {
    if (!convert(thr.selectorTerms[j].podSelector, preq)) tflags[j] |= KT_TERM_POD_SEL_INVALID;
    if (thr.cluster && !convert(thr.selectorTerms[j].namespaceSelector, nreq)) tflags[j] |= KT_TERM_NS_SEL_INVALID;
    tpo[j + 1] = (uint32_t)(preq.val_off.size() - 1);
    tno[j + 1] = (uint32_t)(nreq.val_off.size() - 1);
  }
  uint32_t flags = KT_THR_VALID | (thr.cluster ? KT_THR_CLUSTER : 0u);
  if (thr.throttlerName == p.args.name) flags |= KT_THR_RESPONSIBLE;  // isResponsibleFor (throttle_controller.go:213-215)
  uint32_t ns32 = (uint32_t)ns_row, zero32 = 0, ooff[2] = {0, (uint32_t)no}, toff[2] = {0, (uint32_t)nt};
  uint64_t zero64 = 0;
  DenseAmount empty;
  kt_snapshot b{};
  b.D = D;
  b.n_thr = 1;
  b.thr_flags = &flags;
  b.thr_ns = &ns32;
  b.thr_spec = kt_amounts{spec.v, &spec.present, &spec.count, &spec.has_count};
  b.thr_calc = kt_amounts{empty.v, &empty.present, &empty.count, &empty.has_count};
  b.thr_used = b.thr_calc;
  b.thr_reserved = b.thr_calc;
  b.thr_thrl_flag = &zero32;
  b.thr_thrl_has = &zero32;
  b.thr_status_msgs_fp = &zero64;
  b.thr_spec_msgs_fp = &spec_fp;
  b.thr_ovr_off = ooff;
  b.ovr_begin_s = ob.data();
  b.ovr_begin_ns = obn.data();
  b.ovr_end_s = oe.data();
  b.ovr_end_ns = oen.data();
  b.ovr_flags = oflags.data();
  b.ovr_thr = kt_amounts{ov.data(), opresent.data(), ocount.data(), ohas.data()};
  b.thr_term_off = toff;
  b.term_flags = tflags.data();
  b.term_preq_off = tpo.data();
  b.term_nreq_off = tno.data();
  b.preq = preq.view();
  b.nreq = nreq.view();
  const int32_t r32 = (int32_t)row;
  // The row's stored status starts empty (calculatedAt zero => CheckThrottledFor falls back to spec.threshold,
  // throttle_types.go:129-132) until the next ReconcileAll — the reference enqueues a reconcile for the
  // throttle on the very same Add/Update event (throttle_controller.go:401-417).
  (void)existed;
  int32_t rc = kt_upsert_throttles(p.e, &b, &r32);
  if (rc != KT_OK) { if (err) *err = p.engine_error(rc); return false; }
  p.thr_by_row[(size_t)row] = thr;
  p.thr_live[(size_t)row] = 1;
  p.thr_msgs[(size_t)row] = msgs;
  p.written.erase(tk);
  if (!p.push_reserved(r32, err)) return false;
  return true;
}

bool KubeThrottler::OnThrottleDelete(const std::string& key, bool cluster, std::string* err) {
  std::lock_guard<std::recursive_mutex> lk(p_->mu);
  auto& p = *p_;
  const std::string tk = (cluster ? "C:" : "T:") + key;
  const int64_t row = p.thr_rows.find(tk);
  if (row < 0) return true;
  const int32_t r32 = (int32_t)row;
  int32_t rc = kt_delete_throttles(p.e, 1, &r32);
  if (rc != KT_OK) { if (err) *err = p.engine_error(rc); return false; }
  p.thr_rows.release(tk);
  p.thr_live[(size_t)row] = 0;
  p.reserved.erase(r32);
  p.written.erase(tk);
  return true;
}

// ---------------------------------------------------------------------------------------------------
// PreFilter / Reserve / Unreserve
// ---------------------------------------------------------------------------------------------------
namespace {
const char* status_name(uint8_t s) {
  switch (s) {
    case KT_STATUS_NOT_THROTTLED: return "not-throttled";
    case KT_STATUS_ACTIVE: return "active";
    case KT_STATUS_INSUFFICIENT: return "insufficient";
    case KT_STATUS_POD_REQUESTS_EXCEEDS_THRESHOLD: return "pod-requests-exceeds-threshold";
    case KT_STATUS_ERROR: return "error";
    default: return "";
  }
}
}  // namespace

// row_always: the caller needs the status row even of an allowed pod (Reserve: which throttles the pod affects)
static bool check_one(KubeThrottler::Impl& p, const Pod& pod, KubeThrottler* self, std::vector<uint8_t>* row_out,
                      uint64_t* summary, std::string* err, bool row_always = true) {
  if (!self->OnPodAdd(pod, err)) return false;  // the engine evaluates what the informer cache would hold
  const int64_t row = p.pod_rows.find(pod.Key());
  int32_t T = 0;
  kt_throttle_rows(p.e, &T);
  row_out->assign((size_t)std::max(T, 1), 0);
  // the verdict first: one pod, summary word only = the engine's few-pod path (no copy, no stream synchronisation, not
  // queued behind a running reconcile); the status row is only needed to word the reasons of a pod that is not allowed
  int32_t rc = row_always ? KT_OK : kt_check(p.e, 1, &row, /*isThrottledOnEqual=*/0, summary, nullptr);
  p.last_row_pending = -1;
  if (rc == KT_OK && (row_always || *summary != 0))  // blocked or Error: launch + fetch of the row under ONE engine lock
    rc = kt_check(p.e, 1, &row, /*isThrottledOnEqual=*/0, summary, row_out->data());
  else if (rc == KT_OK && row_out == &p.last_status)
    p.last_row_pending = row;
  if (rc != KT_OK) {
    if (err) *err = p.engine_error(rc);
    return false;
  }
  row_out->resize((size_t)T);
  return true;
}

// reasons in the reference's fixed order (plugin.go:182-214)
static std::vector<std::string> block_reasons(const KubeThrottler::Impl& p, const uint8_t* row, size_t n) {
  std::vector<std::string> out;
  static const uint8_t kOrder[] = {KT_STATUS_POD_REQUESTS_EXCEEDS_THRESHOLD, KT_STATUS_ACTIVE, KT_STATUS_INSUFFICIENT};
  for (uint8_t code : kOrder)
    for (int cluster = 1; cluster >= 0; --cluster) {
      std::string names;
      for (size_t t = 0; t < n; ++t) {
        if (!p.thr_live[t] || p.thr_by_row[t].cluster != (cluster == 1) || row[t] != code) continue;
        if (!names.empty()) names += ",";
        names += p.thr_by_row[t].Key();
      }
      if (!names.empty())
        out.push_back(std::string(cluster ? "clusterthrottle[" : "throttle[") + status_name(code) + "]=" + names);
    }
  return out;
}

// the Warning event PreFilter records when the pod's own requests exceed a threshold (plugin.go:190-202):
// ClusterThrottle names first, then Throttle names
static std::vector<Event> block_events(const KubeThrottler::Impl& p, const uint8_t* row, size_t n) {
  std::string names;
  for (int cluster = 1; cluster >= 0; --cluster)
    for (size_t t = 0; t < n; ++t) {
      if (!p.thr_live[t] || p.thr_by_row[t].cluster != (cluster == 1) || row[t] != KT_STATUS_POD_REQUESTS_EXCEEDS_THRESHOLD) continue;
      if (!names.empty()) names += ",";
      names += p.thr_by_row[t].Key();
    }
  if (names.empty()) return {};
  return {Event{"Warning", "ResourceRequestsExceedsThrottleThreshold",
                "It won't be scheduled unless decreasing resource requests or increasing ClusterThrottle/Throttle threshold "
                "because its resource requests exceeds their thresholds: " + names}};
}

Status KubeThrottler::PreFilter(const Pod& pod) {
  std::lock_guard<std::recursive_mutex> lk(p_->mu);
  auto& p = *p_;
  Status st;
  uint64_t summary = 0;
  std::string err;
  if (!check_one(p, pod, this, &p.last_status, &summary, &err, /*row_always=*/false)) {
    st.code = Error;
    st.reasons.push_back(err);
    return st;
  }
  if (KT_SUMMARY_VERDICT(summary) == KT_VERDICT_ERROR) {  // plugin.go:154-156,166-168
    st.code = Error;
    st.reasons.push_back("throttle check failed for pod " + pod.Key() + " (invalid selector or unknown namespace)");
    return st;
  }
  if (KT_SUMMARY_VERDICT(summary) == KT_VERDICT_SUCCESS) return st;  // plugin.go:177-180
  st.code = UnschedulableAndUnresolvable;
  st.reasons = block_reasons(p, p.last_status.data(), p.last_status.size());
  st.events = block_events(p, p.last_status.data(), p.last_status.size());
  return st;
}

std::string KubeThrottler::LastStatusOf(const std::string& throttle_key) const {
  std::lock_guard<std::recursive_mutex> lk(p_->mu);
  auto& p = *p_;
  if (p.last_row_pending >= 0) {  // the last PreFilter allowed the pod on the summary word alone: fetch its row now
    uint64_t summary = 0;
    int64_t row = p.last_row_pending;
    int32_t T = 0;
    kt_throttle_rows(p.e, &T);
    p.last_status.assign((size_t)std::max(T, 1), 0);
    if (kt_check(p.e, 1, &row, 0, &summary, p.last_status.data()) != KT_OK) p.last_status.assign(p.last_status.size(), 0);
    p.last_status.resize((size_t)T);
    p.last_row_pending = -1;
  }
  for (size_t t = 0; t < p.last_status.size(); ++t)
    if (p.thr_live[t] && p.thr_by_row[t].Key() == throttle_key) return status_name(p.last_status[t]);
  return "";
}

Status KubeThrottler::Reserve(const Pod& pod) {
  std::lock_guard<std::recursive_mutex> lk(p_->mu);
  auto& p = *p_;
  Status st;
  std::vector<uint8_t> row;
  uint64_t summary = 0;
  std::string err;
  if (!check_one(p, pod, this, &row, &summary, &err, /*row_always=*/true) || KT_SUMMARY_VERDICT(summary) == KT_VERDICT_ERROR) {
    st.code = Error;  // plugin.go:223-233
    st.reasons.push_back("Failed to reserve pod=" + pod.Key() + (err.empty() ? "" : ": " + err));
    return st;
  }
  // ResourceAmountOfPod as the engine holds it
  DenseAmount amt;
  const int64_t prow = p.pod_rows.find(pod.Key());
  kt_fetch_pod_requests(p.e, 1, &prow, amt.v, &amt.present);
  amt.has_count = 1;
  amt.count = 1;
  for (size_t t = 0; t < row.size(); ++t) {
    if (row[t] == KT_STATUS_NOT_AFFECTED) continue;  // affectedThrottles (throttle_controller.go:271-292)
    p.reserved[(int32_t)t][pod.Key()] = amt;
    if (!p.push_reserved((int32_t)t, &err)) {
      st.code = Error;
      st.reasons.push_back(err);
      return st;
    }
  }
  return st;
}

void KubeThrottler::Unreserve(const Pod& pod) {
  std::lock_guard<std::recursive_mutex> lk(p_->mu);
  auto& p = *p_;
  for (auto& kv : p.reserved)
    if (kv.second.erase(pod.Key())) p.push_reserved(kv.first, nullptr);
}

bool KubeThrottler::OnPodUpdate(const Pod& old_pod, const Pod& new_pod, std::string* err) {
  std::lock_guard<std::recursive_mutex> lk(p_->mu);
  auto& p = *p_;
  auto counts_in = [&](const Pod& q) { return q.schedulerName == p.args.targetSchedulerName && !q.nodeName.empty(); };
  if (old_pod.Key() != new_pod.Key() || (!counts_in(old_pod) && !counts_in(new_pod))) return OnPodAdd(new_pod, err);
  // affectedThrottles(oldPod) and affectedThrottles(newPod): an error on either side skips the move (:459-468)
  std::vector<uint8_t> before, after;
  uint64_t s_before = 0, s_after = 0;
  std::string ignored;
  const bool ok_before = check_one(p, old_pod, this, &before, &s_before, &ignored) && KT_SUMMARY_VERDICT(s_before) != KT_VERDICT_ERROR;
  if (!OnPodAdd(new_pod, err)) return false;  // from here on the engine holds the new object
  const bool ok_after = check_one(p, new_pod, this, &after, &s_after, &ignored) && KT_SUMMARY_VERDICT(s_after) != KT_VERDICT_ERROR;
  if (!ok_before || !ok_after) return true;
  DenseAmount amt;  // ResourceAmountOfPod(newPod), as the engine computed it
  const int64_t prow = p.pod_rows.find(new_pod.Key());
  int32_t rc = kt_fetch_pod_requests(p.e, 1, &prow, amt.v, &amt.present);
  if (rc != KT_OK) { if (err) *err = p.engine_error(rc); return false; }
  amt.has_count = 1;
  amt.count = 1;
  const size_t n = std::max(before.size(), after.size());
  for (size_t t = 0; t < n; ++t) {
    const bool was = t < before.size() && before[t] != KT_STATUS_NOT_AFFECTED;
    const bool is = t < after.size() && after[t] != KT_STATUS_NOT_AFFECTED;
    if (was == is) continue;  // common throttles keep whatever they hold (symmetric difference only)
    if (was) {
      auto it = p.reserved.find((int32_t)t);
      if (it == p.reserved.end() || !it->second.erase(new_pod.Key())) continue;
    } else {
      p.reserved[(int32_t)t][new_pod.Key()] = amt;
    }
    if (!p.push_reserved((int32_t)t, err)) return false;
  }
  return true;
}

// One scheduling pass over a queue of pending pods IN ORDER: PreFilter, and on Success Reserve — a single engine
// launch (kt_admit_launch) instead of 2 x n calls.  The reserved cache is updated exactly as n
// Reserve calls would have (reserved_resource_amounts.go:66-77), so Unreserve keeps working pod by pod.
std::vector<Status> KubeThrottler::AdmitQueue(const std::vector<std::string>& pod_keys) {
  std::lock_guard<std::recursive_mutex> lk(p_->mu);
  auto& p = *p_;
  const size_t n = pod_keys.size();
  std::vector<Status> out(n);
  std::vector<int64_t> rows(n);
  for (size_t i = 0; i < n; ++i) {
    rows[i] = p.pod_rows.find(pod_keys[i]);
    if (rows[i] < 0) {
      for (auto& st : out) st.code = Error, st.reasons = {"pod " + pod_keys[i] + " is not known to the plugin (OnPodAdd first)"};
      return out;
    }
  }
  if (n == 0) return out;
  // The reservation cache is a map keyed by pod (reserved_resource_amounts.go:130-135: c[nn] = amount, idempotent): a pod
  // that is ALREADY reserved somewhere, or named twice in the queue, must not add its amount a second time — neither to
  // the totals nor to what the following pods are checked against.  The engine's queue walk adds every admitted pod, so
  // such a pod ends the current engine segment and goes through the plain PreFilter + Reserve calls; everything else
  // runs as ONE launch per segment (the whole queue in the common case).
  auto already_reserved = [&](const std::string& key) {
    for (auto& kv : p.reserved)
      if (kv.second.count(key)) return true;
    return false;
  };
  auto admit_segment = [&](size_t i0, size_t i1) {
    const size_t m = i1 - i0;
    if (m == 0) return;
    int32_t T = 0;
    kt_throttle_rows(p.e, &T);
    std::vector<uint64_t> summary(m);
    std::vector<uint8_t> status(m * (size_t)(T > 0 ? T : 1));
    std::vector<int64_t> req(m * (size_t)p.D);
    std::vector<uint32_t> present(m);
    int32_t rc = kt_admit_launch(p.e, (int64_t)m, rows.data() + i0, /*isThrottledOnEqual=*/0, KT_ADMIT_COMMIT, nullptr);
    if (rc == KT_OK) rc = kt_check_fetch(p.e, (int64_t)m, summary.data(), T > 0 ? status.data() : nullptr);
    if (rc == KT_OK) rc = kt_fetch_pod_requests(p.e, (int64_t)m, rows.data() + i0, req.data(), present.data());
    if (rc != KT_OK) {
      for (size_t i = i0; i < i1; ++i) out[i].code = Error, out[i].reasons = {p.engine_error(rc)};
      return;
    }
    for (size_t j = 0; j < m; ++j) {
      const size_t i = i0 + j;
      const uint8_t* row = status.data() + j * (size_t)T;
      const uint64_t v = KT_SUMMARY_VERDICT(summary[j]);
      if (v == KT_VERDICT_ERROR) {
        out[i].code = Error;
        out[i].reasons.push_back("throttle check failed for pod " + pod_keys[i] + " (invalid selector or unknown namespace)");
      } else if (v != KT_VERDICT_SUCCESS) {
        out[i].code = UnschedulableAndUnresolvable;
        out[i].reasons = block_reasons(p, row, (size_t)T);
        out[i].events = block_events(p, row, (size_t)T);
      } else {
        DenseAmount amt;
        for (int d = 0; d < p.D; ++d) amt.v[d] = req[j * (size_t)p.D + d];
        amt.present = present[j];
        amt.has_count = 1;
        amt.count = 1;
        for (int32_t t = 0; t < T; ++t)
          if (row[t] != KT_STATUS_NOT_AFFECTED) p.reserved[t][pod_keys[i]] = amt;  // the engine already holds the totals
      }
    }
  };
  size_t seg0 = 0;
  std::set<std::string> in_segment;
  for (size_t i = 0; i < n; ++i) {
    if (!in_segment.count(pod_keys[i]) && !already_reserved(pod_keys[i])) {
      in_segment.insert(pod_keys[i]);
      continue;
    }
    admit_segment(seg0, i);
    const Pod pod = p.pods.at(pod_keys[i]);
    out[i] = PreFilter(pod);
    if (out[i].IsSuccess()) out[i] = Reserve(pod);
    seg0 = i + 1;
    in_segment.clear();
  }
  admit_segment(seg0, n);
  return out;
}

// ---------------------------------------------------------------------------------------------------
// reconcile
// ---------------------------------------------------------------------------------------------------
bool KubeThrottler::ReconcileAll(const std::string& now_rfc3339, std::map<std::string, ThrottleStatus>* out,
                                 std::string* err) {
  std::lock_guard<std::recursive_mutex> lk(p_->mu);
  auto& p = *p_;
  int64_t now_s;
  int32_t now_ns;
  if (!ParseRFC3339(now_rfc3339, &now_s, &now_ns, err)) return false;
  int32_t T = 0;
  kt_throttle_rows(p.e, &T);
  const int D = p.D;
  const size_t N = (size_t)std::max(T, 1);
  std::vector<int64_t> uv(N * D), ucount(N), cv(N * D), ccount(N);
  std::vector<uint32_t> upresent(N), cpresent(N), tflag(N), thas(N);
  std::vector<uint8_t> uhas(N), chas(N), updated(N), tpod(N), terr(N);
  kt_status st{};
  st.used = kt_amounts{uv.data(), upresent.data(), ucount.data(), uhas.data()};
  st.calc = kt_amounts{cv.data(), cpresent.data(), ccount.data(), chas.data()};
  st.calc_at_nonzero = updated.data();
  st.thrl_flag = tflag.data();
  st.thrl_has = thas.data();
  st.thrl_pod = tpod.data();
  st.error = terr.data();
  int32_t rc = kt_reconcile_launch(p.e, now_s, now_ns, KT_RECONCILE_APPLY, nullptr);
  if (rc == KT_OK) rc = kt_reconcile_fetch(p.e, T, &st);
  std::vector<int64_t> nx_s((size_t)T + 1);
  std::vector<int32_t> nx_ns((size_t)T + 1);
  std::vector<uint8_t> nx_has((size_t)T + 1);
  if (rc == KT_OK) rc = kt_reconcile_fetch_next_override(p.e, T, nx_s.data(), nx_ns.data(), nx_has.data());
  if (rc != KT_OK) {
    if (err) *err = p.engine_error(rc);
    return false;
  }
  // Once status is updated, the reconciled throttle's AFFECTED pods are safe to un-reserve (unreserveAffectedPods,
  // throttle_controller.go:135-155 / clusterthrottle_controller.go:138-160: it walks affectedNonTerminatedPods +
  // affectedTerminatedPods of the throttle — shouldCountIn AND the selector matches the pod as it is now).  A pod that was
  // reserved on the throttle and whose labels moved on afterwards is NOT in that list: its reservation stays until the pod
  // handlers move it (kt_affected_pods: one small check launch over the pods that hold reservations).
  {
    std::vector<int32_t> thr_list;   // reconciled throttles that hold reservations
    std::vector<std::string> keys;   // counted pods that hold a reservation on one of them
    std::map<std::string, size_t> key_ix;
    for (auto& kv : p.reserved) {
      // a reconcile that failed returned before unreserveAffectedPods (throttle_controller.go:103-106), and throttles of
      // another throttler are never reconciled: their reservations stay
      const int32_t t = kv.first;
      if (t < 0 || t >= T || terr[(size_t)t] || !p.thr_live[(size_t)t] || p.thr_by_row[(size_t)t].throttlerName != p.args.name) continue;
      bool any = false;
      for (auto& r : kv.second) {
        auto pit = p.pods.find(r.first);
        const bool counted = pit != p.pods.end() && pit->second.schedulerName == p.args.targetSchedulerName && !pit->second.nodeName.empty();
        if (!counted || p.pod_rows.find(r.first) < 0) continue;
        if (!key_ix.count(r.first)) key_ix[r.first] = keys.size(), keys.push_back(r.first);
        any = true;
      }
      if (any) thr_list.push_back(t);
    }
    if (!keys.empty() && !thr_list.empty()) {
      std::vector<int64_t> rows(keys.size());
      for (size_t i = 0; i < keys.size(); ++i) rows[i] = p.pod_rows.find(keys[i]);
      std::vector<uint8_t> aff(keys.size() * thr_list.size());
      rc = kt_affected_pods(p.e, (int64_t)rows.size(), rows.data(), (int32_t)thr_list.size(), thr_list.data(), aff.data());
      if (rc != KT_OK) {
        if (err) *err = p.engine_error(rc);
        return false;
      }
      for (size_t j = 0; j < thr_list.size(); ++j) {
        auto& held = p.reserved[thr_list[j]];
        bool changed = false;
        for (auto it = held.begin(); it != held.end();) {
          auto k = key_ix.find(it->first);
          if (k != key_ix.end() && aff[k->second * thr_list.size() + j] == 1) it = held.erase(it), changed = true;
          else ++it;
        }
        if (changed) p.push_reserved(thr_list[j], nullptr);
      }
    }
  }
  std::vector<std::string> dim_name((size_t)D);
  std::vector<int> dim_scale((size_t)D, 0);
  for (auto& kv : p.dims) dim_name[(size_t)kv.second.first] = kv.first, dim_scale[(size_t)kv.second.first] = kv.second.second;
  for (int32_t t = 0; t < T; ++t) {
    if (!p.thr_live[(size_t)t] || p.thr_by_row[(size_t)t].throttlerName != p.args.name) continue;
    ThrottleStatus s;
    s.error = terr[t] != 0;
    s.usedHasCounts = uhas[t] != 0;
    s.usedPod = ucount[t];
    s.throttledPod = tpod[t] != 0;
    s.calculatedThresholdUpdated = updated[t] != 0;
    s.messages = p.thr_msgs[(size_t)t];
    s.hasNextOverride = nx_has[(size_t)t] != 0;  // NextOverrideHappensIn -> enqueueAfter (throttle_controller.go:201-208)
    s.nextOverrideSec = nx_s[(size_t)t];
    s.nextOverrideNsec = nx_ns[(size_t)t];
    for (int d = 0; d < D; ++d) {
      if ((upresent[t] >> d) & 1u) {
        Quantity q;
        q.nano = (__int128)uv[(size_t)t * D + d];
        for (int k = 0; k < 9 + dim_scale[d]; ++k) q.nano *= 10;
        auto f = p.dim_format.find(dim_name[d]);
        if (f != p.dim_format.end()) q.format = f->second;
        s.used[dim_name[d]] = q;
      }
      if ((thas[t] >> d) & 1u) s.throttledRequests[dim_name[d]] = (tflag[t] >> d) & 1u;
    }
    if (!s.error) {  // a reconcile error returns before any status is built (throttle_controller.go:103-106)
      ThrottleStatus& prev = p.written[Impl::thr_key(p.thr_by_row[(size_t)t])];
      s.needsUpdate = s.calculatedThresholdUpdated || !StatusSemanticEqual(prev, s);
      prev = s;
    }
    if (out) (*out)[p.thr_by_row[(size_t)t].Key()] = s;
  }
  return true;
}

}  // namespace kth
