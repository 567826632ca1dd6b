#!/bin/bash
# AddressSanitizer + UBSan run of the CPU-side code (host library and oracle) under the -m "not gpu" tests that
# exercise them.  CPU box only (GPU sanitizers are not available on the pool).  Usage: tools/run_asan.sh
set -e
cd "$(dirname "$0")/.."
make -s -C mcmc-ammsb-gpu_amd/csrc
make -s -C mcmc-ammsb-gpu_amd/host asan
make -s -C oracle asan
export AMMSB_HOST_LIB=$PWD/mcmc-ammsb-gpu_amd/libammsb_host_asan.so
export AMMSB_ORACLE_LIB=$PWD/oracle/libammsb_oracle_asan.so
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export OMP_NUM_THREADS=2
python -m pytest -x -q -m "not gpu" tests/test_host_lib.py tests/test_oracle_samplers.py tests/test_oracle_pins.py \
    tests/test_checkpoint.py tests/test_oracle_model.py tests/test_golden.py -p no:cacheprovider "$@"
# the command-line driver (flag parser, SNAP / gzip data-set loaders) and the exchange rendezvous (sockets): the
# executables themselves are sanitizer builds (make asan), started by the CPU cases of their tests
unset LD_PRELOAD AMMSB_HOST_LIB AMMSB_ORACLE_LIB
export AMMSB_MAIN_EXE=$PWD/mcmc-ammsb-gpu_amd/ammsb_main_asan
export AMMSB_XT_EXE=$PWD/mcmc-ammsb-gpu_amd/exchange_test_asan
python -m pytest -x -q -m "not gpu" tests/test_cli.py tests/test_cpp_exchange.py -p no:cacheprovider "$@"
