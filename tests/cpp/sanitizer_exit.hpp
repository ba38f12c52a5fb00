// sanitizer_exit.hpp -- how the C++ test programs leave under AddressSanitizer.
//
// With the round-4 library (workspaces own non-blocking HIP streams, created and destroyed with the planners) the HIP / HSA
// runtime's OWN finalizer, run from __cxa_finalize at process exit, frees memory after ASan's device allocator has marked the
// device runtime as unloaded -- "CHECK failed: sanitizer_allocator_device.h:125 dev_runtime_unloaded_" -- inside
// libhsa-runtime64.so, after main() has returned and before stdio is flushed (profiles/r04_asan_host_pass.log, first
// attempt).  Nothing of this repository is on that stack.  An ASan build therefore runs the leak check itself, flushes its
// output and leaves with _exit(): every check of the program and LeakSanitizer's report still happen, the runtime's teardown
// does not.
#pragma once
#include <cstdio>
#include <cstdlib>
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define PHAST_TEST_UNDER_ASAN 1
#endif
#endif
#ifdef PHAST_TEST_UNDER_ASAN
#include <sanitizer/lsan_interface.h>
#include <unistd.h>
#endif

[[noreturn]] inline void phast_test_exit(int code) {
    std::fflush(stdout);
#ifdef PHAST_TEST_UNDER_ASAN
    if (__lsan_do_recoverable_leak_check() != 0 && code == 0) code = 23;  // what ASAN_OPTIONS=detect_leaks=1 would report at exit
    std::fflush(stdout);
    _exit(code);
#else
    std::exit(code);
#endif
}
