#pragma once
#include <cstdlib>

// Developer A/B switches (environment variables read inside the schedules: "keep round 4's passes", "one stream", ...).  The PRODUCT library is built
// without them -- cerb_dev_getenv() is then a constant nullptr and every switch folds to its default at compile time; cerberus_amd/build.py also
// builds libcerberus_hip_dev.so with -DCERB_DEV_SWITCHES, which the A/B tests load in a child process (tests/conftest.py: dev_switches).
static inline const char* cerb_dev_getenv(const char* name) {
#ifdef CERB_DEV_SWITCHES
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
