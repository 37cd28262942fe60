H=tools/fasn_harness
run() { echo "--- $*"; for lib in old new old new; do if [ $lib = old ]; then export LD_LIBRARY_PATH=$PWD/tools/old; else unset LD_LIBRARY_PATH; fi; echo -n "$lib: "; $H bench "$@" 2>&1 | grep bwd; done; unset LD_LIBRARY_PATH; }
run 8 16 4096 4096 64 1 0 0 100 1
run 8 16 4096 4096 64 0 1 0 100 1
run 8 16 1024 1024 64 1 0 0 300 1
run 4 32 8192 8192 128 1 0 0 10 1
run 4 32 8192 8192 128 1 1 0 10 1
run 8 16 4096 4096 32 1 0 0 50 1
run 8 16 4096 4096 64 1 0 0 50 1 1 2 0
run 4 32 8192 8192 128 1 0 0 5 1 0.5 2 1
