// bicg_knobs.h -- the library's run-time switches come in three kinds.
//   getenv("BICG_...") directly: switches for users (tabled in INTEGRATION.md section 6).
//   knob_tok(set, "name"):       the forms of the plan and the hooks of the tests, gathered in three variables that hold
//                                comma-separated `name` / `name=value` tokens: BICG_PLAN (which layouts and products the plan may
//                                use: "stencil=0,lines=2,planes=8"), BICG_PERSIST ("0", or "chunk=64,shifted=0") and BICG_TEST
//                                (fault injection and forced paths: "force-comm,spin-ticks=0"). Every form gives the same bits or
//                                is covered by a test that says what differs; the table is in INTEGRATION.md section 6.
//   knob_x("BICG_..."):          measurement knobs of the development rounds -- A/B settings whose outcome is on record in
//                                profiles/NOTES.md, negative results kept for reference. They are read only by a library built
//                                with `make EXPERIMENTS=1` (-DBICG_EXPERIMENTS); the default build has their defaults compiled in.
#pragma once

#include <cstdlib>
#include <cstring>

namespace bicg {

// Value of token `name` in the comma-separated list $set ("1" for a bare token), nullptr when the variable or the token is
// absent. The text lives in one of four per-thread buffers, so a caller may hold a few values at a time.
inline const char *knob_tok(const char *set, const char *name)
{
    const char *s = getenv(set);
    if (!s) return nullptr;
    static thread_local char buf[4][48];
    static thread_local unsigned turn = 0;
    const size_t ln = strlen(name);
    while (*s) {
        while (*s == ',' || *s == ' ') ++s;
        const char *e = s;
        while (*e && *e != ',' && *e != ' ') ++e;
        if ((size_t)(e - s) >= ln && !strncmp(s, name, ln) && (s + ln == e || s[ln] == '=')) {
            char *out = buf[turn++ & 3];
            if (s + ln == e) { out[0] = '1'; out[1] = 0; return out; }
            size_t lv = (size_t)(e - s) - ln - 1;
            if (lv > sizeof(buf[0]) - 1) lv = sizeof(buf[0]) - 1;
            memcpy(out, s + ln + 1, lv);
            out[lv] = 0;
            // switches may be spelled out: on / true / yes read as 1, off / false / no as 0 (callers use atoi)
            for (const char *w : {"on", "true", "yes"}) if (!strcmp(out, w)) { out[0] = '1'; out[1] = 0; }
            for (const char *w : {"off", "false", "no"}) if (!strcmp(out, w)) { out[0] = '0'; out[1] = 0; }
            return out;
        }
        s = e;
    }
    return nullptr;
}
// The names each list knows. A token that is none of them is a typing error that would otherwise
// pass silently -- "stencil=O" selects nothing --: knob_unknown copies the first such token to out and returns how many there are.
// A known name with a value that cannot be read counts too: values are integers or on / off / true / false / yes / no ("layout":
// jag or pad), so "lines=2;planes=8" (one token whose value is "2;planes=8") and "stencil = 0" (three tokens) are reported.
inline bool knob_value_ok(const char *name, size_t name_len, const char *v, size_t lv)
{
    auto is = [&](const char *w) { return strlen(w) == lv && !strncmp(v, w, lv); };
    if (name_len == 6 && !strncmp(name, "layout", 6)) return is("jag") || is("pad");
    if (is("on") || is("off") || is("true") || is("false") || is("yes") || is("no")) return true;
    if (lv == 0) return false;
    size_t i = (v[0] == '-' || v[0] == '+') ? 1 : 0;
    if (i == lv) return false;
    for (; i < lv; ++i) if (v[i] < '0' || v[i] > '9') return false;
    return true;
}
inline const char *const *knob_names(const char *set)
{
    static const char *const plan[] = {"stencil", "lines", "planes", "ca-fuse", "layout", "window", "col16", "uniform", "constant", "masked",
                                       "desc", "lists", "jagw", "spmm", "spmm-window", "fuse-pipe", "pipe-probe", "halo-fused", "window-list", "wide", nullptr};
    static const char *const persist[] = {"0", "off", "chunk", "shifted", nullptr};      // (the persistent forms are the default: there is no "on")
    static const char *const test[] = {"force-comm", "spin-ticks", "p2p-fault-after", "plan-collide", "spmm-skip", "spmm-gstep", "spmm-tile", "spmm-jres", nullptr};
    static const char *const none[] = {nullptr};
    return !strcmp(set, "BICG_PLAN") ? plan : !strcmp(set, "BICG_PERSIST") ? persist : !strcmp(set, "BICG_TEST") ? test : none;
}
inline int knob_unknown(const char *set, char *out, size_t cap)
{
    const char *s = getenv(set);
    int n = 0;
    if (out && cap) out[0] = 0;
    if (!s) return 0;
    const char *const *names = knob_names(set);
    while (*s) {
        while (*s == ',' || *s == ' ') ++s;
        const char *e = s, *q = s;
        while (*e && *e != ',' && *e != ' ') ++e;
        while (q < e && *q != '=') ++q;
        if (e == s) break;
        bool known = false;
        for (int i = 0; names[i]; ++i) known = known || (strlen(names[i]) == (size_t)(q - s) && !strncmp(names[i], s, (size_t)(q - s)));
        if (known && q < e && !knob_value_ok(s, (size_t)(q - s), q + 1, (size_t)(e - q - 1))) known = false;
        if (!known && n++ == 0 && out && cap) { const size_t l = (size_t)(e - s) < cap - 1 ? (size_t)(e - s) : cap - 1; memcpy(out, s, l); out[l] = 0; }
        s = e;
    }
    return n;
}
inline const char *plan_tok(const char *name) { return knob_tok("BICG_PLAN", name); }
inline const char *test_tok(const char *name) { return knob_tok("BICG_TEST", name); }
inline bool plan_off(const char *name) { const char *v = plan_tok(name); return v && atoi(v) == 0; }     // "name=0"

#ifdef BICG_EXPERIMENTS
inline const char *knob_x(const char *name) { return getenv(name); }
constexpr bool kExperiments = true;
#else
inline const char *knob_x(const char *) { return nullptr; }
constexpr bool kExperiments = false;
#endif

}  // namespace bicg
