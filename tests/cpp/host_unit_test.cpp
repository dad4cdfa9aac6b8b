// host_unit_test — CPU-only checks of the host mirror's pure helpers (no engine call, so no GPU needed):
// Quantity.Add's format rule + Quantity.String(), and the Semantic.DeepEqual stand-in that decides whether a
// reconcile has to call UpdateStatus (throttle_controller.go:157).
#include <cstdio>
#include <memory>
#include <string>

#include "kt_host.hpp"

using namespace kth;

static int g_fail = 0;
#define EXPECT(cond)                                                         \
  do {                                                                       \
    if (!(cond)) {                                                           \
      ++g_fail;                                                              \
      fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);        \
    }                                                                        \
  } while (0)

static Quantity Q(const char* text) {
  Quantity q;
  std::string err;
  EXPECT(ParseQuantity(text, &q, &err));
  return q;
}
static std::string SumOf(std::initializer_list<const char*> texts) {
  Quantity total;
  for (const char* t : texts) AddQuantity(&total, Q(t));
  return FormatQuantity(total);
}

int main() {
  // sums as the reference reports them
  EXPECT(SumOf({"50m", "50m", "50m", "50m", "50m", "50m", "50m", "50m", "50m", "50m", "50m", "50m", "50m", "50m", "50m",
                "50m", "50m", "50m", "50m", "50m"}) == "1");                     // test/integration/throttle_test.go:189
  EXPECT(SumOf({"100m", "100m"}) == "200m");                                     // throttle_test.go:88
  EXPECT(SumOf({"512Mi", "512Mi"}) == "1Gi");
  EXPECT(SumOf({"0", "512Mi", "1G"}) == "1536870912");     // BinarySI kept from the first non-zero addend; no 1024 factor
  EXPECT(SumOf({"1G", "512Mi"}) == "1536870912");          // DecimalSI: no power of ten either
  EXPECT(SumOf({"1e3", "500"}) == "1500" && SumOf({"1e3", "1e3"}) == "2e3");
  EXPECT(SumOf({"1Gi", "-1Gi", "250m"}) == "250m");        // back at zero: the next addend decides again
  EXPECT(FormatQuantity(Q("1.5Gi")) == "1536Mi" && FormatQuantity(Q("0.5Ki")) == "512" && FormatQuantity(Q("1100m")) == "1100m");

  // Semantic.DeepEqual(thr.Status, newStatus)
  ThrottleStatus a, b;
  EXPECT(StatusSemanticEqual(a, b));
  a.used["cpu"] = Q("1");
  EXPECT(!StatusSemanticEqual(a, b) && !StatusSemanticEqual(b, a));
  b.used["cpu"] = Q("1000m");                              // same value, other spelling
  EXPECT(StatusSemanticEqual(a, b));
  b.used["cpu"] = Q("1001m");
  EXPECT(!StatusSemanticEqual(a, b));
  b.used["cpu"] = Q("1");
  b.usedHasCounts = true;                                  // resourceCounts: nil vs &{Pod:0}
  EXPECT(!StatusSemanticEqual(a, b));
  a.usedHasCounts = true, a.usedPod = 2, b.usedPod = 2;
  EXPECT(StatusSemanticEqual(a, b));
  b.usedPod = 3;
  EXPECT(!StatusSemanticEqual(a, b));
  b.usedPod = 2;
  a.throttledRequests["cpu"] = false;                      // map{cpu:false} vs nil map differ; {cpu:false} vs {cpu:true} differ
  EXPECT(!StatusSemanticEqual(a, b));
  b.throttledRequests["cpu"] = true;
  EXPECT(!StatusSemanticEqual(a, b));
  b.throttledRequests["cpu"] = false;
  EXPECT(StatusSemanticEqual(a, b));
  b.throttledPod = true;
  EXPECT(!StatusSemanticEqual(a, b));
  b.throttledPod = false;
  b.hasNextOverride = true, b.messages = {"x"}, b.needsUpdate = true;   // not part of the comparison
  EXPECT(StatusSemanticEqual(a, b));
  auto text = a.UsedStrings();
  EXPECT(text.size() == 1 && text["cpu"] == "1");

  // NewPlugin: DecodePluginArgs' checks come first (plugin_args.go:46-51), with the reference's messages
  {
    std::string err;
    PluginArgs a;
    EXPECT(NewPlugin(a, &err) == nullptr && err == "Name must not be empty");
    a.name = "kube-throttler";
    EXPECT(NewPlugin(a, &err) == nullptr && err == "TargetSchedulerName must not be empty");
    EXPECT(std::string(PluginName) == "kube-throttler");                                    // plugin.go:45
    // the engine has no CPU fallback: on a box without a GPU the plugin cannot be created, and says why
    a.targetSchedulerName = "my-scheduler";
    std::unique_ptr<KubeThrottler> k = NewPlugin(a, &err);
    if (!k) EXPECT(err.find("kt_engine_create") != std::string::npos);
  }

  if (g_fail) {
    fprintf(stderr, "%d expectation(s) failed\n", g_fail);
    return 1;
  }
  printf("host_unit_test: all expectations held\n");
  return 0;
}
