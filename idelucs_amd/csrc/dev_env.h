// dev_env.h -- ONE environment variable carries every diagnostic / tuning knob of the library:
//     IDELUCS_DEV="name=value,name=value,..."        (a bare name reads as "1")
// The Python side takes its own keys (fused.VARIANTS, posthoc.OPTIONS, utils.OPTIONS) from the same variable (_lib.DEV) and ignores the rest, as this side
// ignores the Python keys.  Read at every use (tests change it between launches); the returned string lives in a small per-thread ring.
// The environment variables that remain variables of their own are the user-facing ones: DESIGN.md 5.4 has the table.
#pragma once
#include <cstdlib>
#include <cstring>

namespace idl {

inline const char *dev_env(const char *name)
{
    static thread_local char ring[4][64];
    static thread_local unsigned at = 0;
    const char *s = getenv("IDELUCS_DEV");
    if (s == nullptr) return nullptr;
    const size_t n = strlen(name);
    while (*s) {
        const char *e = strchr(s, ',');
        const size_t len = e ? (size_t)(e - s) : strlen(s);
        if (len >= n && strncmp(s, name, n) == 0 && (len == n || s[n] == '=')) {
            char *b = ring[at++ & 3u];
            if (len == n) { b[0] = '1'; b[1] = 0; return b; }
            size_t vl = len - n - 1;
            if (vl > 63) vl = 63;
            memcpy(b, s + n + 1, vl);
            b[vl] = 0;
            return b;
        }
        s += len;
        if (*s == ',') ++s;
    }
    return nullptr;
}

}  // namespace idl
