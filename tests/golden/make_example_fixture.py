"""Regenerates tests/golden/example_config0.json from the reference's example manifests.

BASELINE.json configs[0] is "example/throttle.yaml + pod1/pod2/pod3 on the reference Go CPU path".
/root/reference does not exist on the GPU box, so the manifests' *content* (names, labels, requests,
threshold) is captured once, here, into a small JSON fixture.  Run in the build container only:

    python tests/golden/make_example_fixture.py
"""
import json
import os

import yaml

REF = "/root/reference/example"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "example_config0.json")


def main():
    out = {"source": "everpeace/kube-throttler example/ (throttle.yaml, clthrottle.yaml, pod1.yaml, pod2.yaml, "
                     "pod3.yaml, pod1m.yaml, throttle-with-temporaryThresholdOverrides.yaml)",
           "manifests": {}}
    for f in ("throttle", "clthrottle", "pod1", "pod2", "pod3", "pod1m", "throttle-with-temporaryThresholdOverrides",
              "clthrottle-with-temporaryThresholdOverrides"):
        with open(os.path.join(REF, f + ".yaml")) as fh:
            m = yaml.load(fh, Loader=yaml.BaseLoader)  # keep timestamps as text
        m.pop("apiVersion", None)
        if m["kind"] == "Pod":
            for c in m["spec"]["containers"]:
                c.pop("image", None)
                c.pop("command", None)
        out["manifests"][f] = m
    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True, default=str)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
