// Every environment variable the library reads goes through dcd_env(): they are A/B-timing and test switches, not configuration
// (each one is listed, with its values and default, in include/dcd_hip.h under "Environment").  A release build that must not
// depend on its process environment defines DCD_NO_TUNING_ENV: dcd_env() then returns NULL for every name and every switch keeps
// its documented default.  tests/test_abi.py checks that the names used here and the header's list are the same set.
#pragma once
#include <stdlib.h>

static inline const char *dcd_env(const char *name)
{
#ifdef DCD_NO_TUNING_ENV
    (void)name;
    return nullptr;
#else
    return getenv(name);
#endif
}
