// msfm_guard.h -- the exception barrier of the C ABI (include/msfm_match.h:11-12: "every call returns an int status, never throws").
//
// The host state behind the entry points is std::vector / std::string / std::map: an allocation failure on a large job's pair tables (or
// any other C++ exception) must not unwind through a C caller -- a cgo / ctypes / JNI frame has no landing pad and the process
// terminates.  Every `extern "C"` function of csrc/msfm_match.hip runs its body through msfm_guard(): std::bad_alloc -> MSFM_E_DEVICE
// ("out of host memory"), any other std::exception -> MSFM_E_INVALID with its what(), anything else -> MSFM_E_DEVICE; the text goes to
// the context's msfm_last_error through `set_error` (which must not throw itself: it is called with a static string or e.what()).
// Pure C++ (no HIP): tests/test_abi.py builds it with g++ and throws through it; the same test checks that no entry point of
// msfm_match.hip is left outside it.
#pragma once
#include <exception>
#include <new>

#ifndef MSFM_E_INVALID
#define MSFM_GUARD_E_INVALID 1
#define MSFM_GUARD_E_DEVICE 2
#else
#define MSFM_GUARD_E_INVALID MSFM_E_INVALID
#define MSFM_GUARD_E_DEVICE MSFM_E_DEVICE
#endif

template <class SetError, class Body>
int msfm_guard(SetError&& set_error, Body&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        set_error(MSFM_GUARD_E_DEVICE, "out of host memory (std::bad_alloc) inside the call");
        return MSFM_GUARD_E_DEVICE;
    } catch (const std::exception& e) {
        set_error(MSFM_GUARD_E_INVALID, e.what());
        return MSFM_GUARD_E_INVALID;
    } catch (...) {
        set_error(MSFM_GUARD_E_DEVICE, "unknown C++ exception inside the call");
        return MSFM_GUARD_E_DEVICE;
    }
}
