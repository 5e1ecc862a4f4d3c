// Drives the host-side mirror of the Pair text interface (csrc/pair_host.hpp) through its grammar, error paths,
// mixing, table hand-over and single() under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only:
// tests/test_sanitizers.py compiles and runs this; exit code 0 and "ok" on stdout = clean).
#include <cstdio>
#include <initializer_list>
#include <string>
#include <vector>

#include "pair_host.hpp"

using polar::PairHost;

static int failures = 0;

static bool run_settings(PairHost &p, std::initializer_list<const char *> a, const char *expect_error) {
  std::vector<const char *> v(a);
  try {
    p.settings((int)v.size(), v.data());
  } catch (const polar::InputError &e) {
    if (!expect_error || std::string(e.what()) != expect_error) {
      std::printf("settings: unexpected error '%s'\n", e.what());
      failures++;
    }
    return false;
  }
  if (expect_error) {
    std::printf("settings: expected '%s', got none\n", expect_error);
    failures++;
  }
  return true;
}

static void run_coeff(PairHost &p, int ntypes, std::initializer_list<const char *> a, bool expect_error) {
  std::vector<const char *> v(a);
  bool threw = false;
  try {
    p.coeff(ntypes, (int)v.size(), v.data());
  } catch (const polar::InputError &) {
    threw = true;
  }
  if (threw != expect_error) {
    std::printf("coeff: error=%d expected=%d\n", (int)threw, (int)expect_error);
    failures++;
  }
}

int main() {
  {
    PairHost p;
    run_settings(p, {}, "Illegal pair_style command");
    run_settings(p, {"9"}, nullptr);
    run_settings(p, {"9", "precision"}, "Expected floating point parameter in input script or data file");  // arg 1 is the Coulomb cutoff
    run_settings(p, {"9", "9", "precision"}, "Illegal pair_style command");
    run_settings(p, {"9", "9", "damp_type", "thole"}, "Illegal pair_style command");
    run_settings(p, {"9", "9", "zodid", "yes"}, "Zodid doesn't work with polar_gs or polar_gs_ranked");
    run_settings(p, {"9", "9", "polar_gs", "yes"}, "polar_gs and polar_gs_ranked are mutually exclusive");
    run_settings(p, {"x9", "9"}, "Expected floating point parameter in input script or data file");
    run_settings(p, {"9", "9", "max_iterations", "3.5"}, "Expected integer parameter in input script or data file");
    run_settings(p, {"2.5", "12.8345", "precision", "1e-11", "max_iterations", "100", "damp_type", "exponential", "damp",
                     "2.1304", "polar_gs_ranked", "yes", "debug", "no", "use_previous", "yes", "dd_cutoff", "12.8345",
                     "device_neigh", "yes", "restart_polar", "yes", "polar_gamma", "1.0", "fixed_iteration", "no"},
                 nullptr);
  }
  {
    PairHost p;
    run_settings(p, {"2.5", "12.0"}, nullptr);
    run_coeff(p, 4, {"1", "1", "0.1"}, true);
    run_coeff(p, 4, {"0", "1", "0.1", "3.0"}, true);
    run_coeff(p, 4, {"1", "9", "0.1", "3.0"}, true);
    run_coeff(p, 4, {"1", "1", "0.10", "3.0", "9.0"}, false);
    run_coeff(p, 4, {"2*3", "2*3", "0.20", "3.5"}, false);
    run_coeff(p, 4, {"*", "4", "0.05", "3.3", "13.0"}, false);
    run_coeff(p, 4, {"4*", "*4", "0.01", "2.0"}, false);
    const char *mod[] = {"mix", "arithmetic", "shift", "yes", "table", "12", "tabinner", "1.5", "tail", "no"};
    p.modify(10, mod);
    double slj[4] = {1, 0, 0, 0.5}, sc[4] = {1, 0, 0, 0.8333};
    p.init(0.21, 332.06371, slj, sc);
    // no tables yet: single() must refuse, not read empty vectors
    double ff = 0;
    bool threw = false;
    try { p.single(0.3, -0.3, 1, 2, 9.0, 1.0, 1.0, ff); } catch (const polar::InputError &) { threw = true; }
    if (!threw) { std::printf("single without tables did not throw\n"); failures++; }
    // a table of the reference's shape (4096 entries); values only need to be finite here
    const int nbits = 12;
    std::vector<double> t[8];
    const double *tp[8];
    for (int k = 0; k < 8; k++) { t[k].assign((size_t)1 << nbits, 0.01 * (k + 1)); tp[k] = t[k].data(); }
    int mask = 0, shift = 0;
    {  // init_bitmap of src/pair.cpp:1619-1683 for (tabinner, cut) = (1.5, 12): only mask/shift matter to the lookup
      shift = 23 - (nbits - 4);
      mask = ((1 << nbits) - 1) << shift;
    }
    p.set_tables(nbits, mask, shift, 2.25, tp);
    double acc = 0;
    for (int i = 1; i <= 4; i++)
      for (int j = 1; j <= 4; j++)
        for (double r = 0.9; r < 13.5; r += 0.37) acc += p.single(0.4, -0.7, i, j, r * r, 0.5, 0.5, ff) + ff;
    if (!(acc == acc)) { std::printf("single produced NaN\n"); failures++; }
    threw = false;
    try { const double *bad[8] = {tp[0], tp[1], nullptr, tp[3], tp[4], tp[5], tp[6], tp[7]}; p.set_tables(nbits, mask, shift, 2.25, bad); }
    catch (const polar::InputError &) { threw = true; }
    if (!threw) { std::printf("null table accepted\n"); failures++; }
    threw = false;
    try { p.set_tables(30, mask, shift, 2.25, tp); } catch (const polar::InputError &) { threw = true; }
    if (!threw) { std::printf("30-bit table accepted\n"); failures++; }
    const char *notab[] = {"table", "0"};   // pair_modify table 0: closed-form branch
    p.modify(2, notab);
    p.set_tables(0, 0, 0, 0.0, tp);
    acc += p.single(0.4, -0.7, 1, 2, 16.0, 1.0, 1.0, ff);
    for (int i = 1; i <= 4; i++) for (int j = i; j <= 4; j++) acc += p.init_one(i, j);
  }
  {
    PairHost q;  // missing coefficients
    run_settings(q, {"2.5", "12.0"}, nullptr);
    run_coeff(q, 2, {"1", "1", "0.1", "3.0"}, false);
    bool threw = false;
    double one[4] = {1, 0, 0, 0};
    try { q.init(0.2, 332.0, one, one); } catch (const polar::InputError &) { threw = true; }
    if (!threw) { std::printf("unset coefficients accepted\n"); failures++; }
  }
  std::printf(failures ? "FAILED %d\n" : "ok\n", failures);
  return failures ? 1 : 0;
}
