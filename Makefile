# Convenience targets; the driver uses __graft_entry__.build() / pytest / bench.py directly.
.PHONY: build test test-gpu bench resources clean
build:
	python -c "import __graft_entry__ as g; g.build()"
test: build
	python -m pytest tests -q -m "not gpu"
test-gpu: build
	python -m pytest tests -q -m gpu
bench: build
	python bench.py --gpus 1
resources:
	tools/kernel_resources.sh
clean:
	$(MAKE) -C kube_throttler_amd/csrc clean
	$(MAKE) -C kube_throttler_amd/host clean
	$(MAKE) -C oracle clean 2>/dev/null || true
