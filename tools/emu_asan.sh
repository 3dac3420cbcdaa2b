#!/bin/bash
# The host-emulation tests with AddressSanitizer on every global / heap access the kernels make (out-of-bounds reads and writes of the
# tensors they are handed show up as ASAN reports naming the kernel source line).  No GPU needed.
#   bash tools/emu_asan.sh [pytest args, default: tests/test_host_emulation_cpu.py -x -q]
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
cd "$(dirname "$0")/.."
export PTC_EMU_ASAN=1 ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:verify_asan_link_order=0:symbolize=1
export ASAN_SYMBOLIZER_PATH=/opt/rocm/lib/llvm/bin/llvm-symbolizer
if [ $# -eq 0 ]; then set -- tests/test_host_emulation_cpu.py -x -q; fi
LD_PRELOAD=$RT python -m pytest "$@"
