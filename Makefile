# Convenience entry points; __graft_entry__.build() does the same from Python.
all: lib oracle cpp

lib:
	$(MAKE) -C minarrow_amd/csrc -j8
oracle:
	$(MAKE) -C oracle
cpp: lib
	$(MAKE) -C tests/cpp
bindings: 
	python3 tools/gen_rust_ffi.py
test-cpu: all
	python3 -m pytest tests -q -m "not gpu"
test-gpu: all
	python3 -m pytest tests -q -m gpu
bench: all
	python3 bench.py
clean:
	$(MAKE) -C minarrow_amd/csrc clean
	$(MAKE) -C tests/cpp clean
	rm -rf oracle/_build oracle/_ref build
.PHONY: all lib oracle cpp bindings test-cpu test-gpu bench clean
