// pair_host.hpp -- host-side mirror of the reference's Pair text interface for this style.
//
// Same grammar, defaults, ordering quirks and error strings as the reference:
//   settings   <- PairLJCutCoulLongPolarization::settings   PS.cpp:678-766
//   coeff      <- ::coeff                                   PS.cpp:772-800
//   modify     <- Pair::modify_params (subset)              src/pair.cpp:125-185
//   init       <- Pair::init / ::init_style / ::init_one    src/pair.cpp:189-263, PS.cpp:806-921
//   tables     <- handed over (polar_set_coul); Pair::init_tables stays LAMMPS host code
//   single     <- ::single                                  PS.cpp:1035-1097
// Pure host C++ (no HIP); the device library consumes the tables it produces.
#pragma once

#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/polar_mi355x.h"

namespace polar {

struct InputError : std::runtime_error {
  explicit InputError(const char *m) : std::runtime_error(m) {}
};

inline void settings_defaults(polar_settings &s) {  // PS.cpp:65-78
  s.cut_lj_global = 0.0;
  s.cut_coul = 0.0;
  s.iterations_max = 50;
  s.damping_type = POLAR_DAMP_NONE;
  s.polar_damp = 2.1304;
  s.zodid = 0;
  s.polar_precision = 0.00000000001;
  s.fixed_iteration = 0;
  s.polar_gs = 0;
  s.polar_gs_ranked = 1;
  s.polar_gamma = 1.03;
  s.use_previous = 0;
  s.debug = 0;
  s.dd_cutoff = 0.0;
  s.device_neigh = 0;
  s.restart_polar = 0;
  s.deterministic = 0;
  s.polar_sor = 1.0;
  s.rccl_halo = 0;
  s.polar_accel = 0;
}

// Force::numeric / Force::inumeric behaviour: whole token must parse.
inline double numeric(const char *str) {
  if (!str || !*str) throw InputError("Expected floating point parameter in input script or data file");
  char *end = nullptr;
  double v = strtod(str, &end);
  if (*end) throw InputError("Expected floating point parameter in input script or data file");
  return v;
}
inline int inumeric(const char *str) {
  if (!str || !*str) throw InputError("Expected integer parameter in input script or data file");
  char *end = nullptr;
  long v = strtol(str, &end, 10);
  if (*end) throw InputError("Expected integer parameter in input script or data file");
  return (int)v;
}
// Force::bounds wildcard grammar
inline void bounds(const char *str, int nmax, int &nlo, int &nhi) {
  const char *star = strchr(str, '*');
  size_t n = strlen(str);
  if (!star) nlo = nhi = atoi(str);
  else if (n == 1) { nlo = 1; nhi = nmax; }
  else if (star == str) { nlo = 1; nhi = atoi(str + 1); }
  else if (star == str + n - 1) { nlo = atoi(str); nhi = nmax; }
  else { nlo = atoi(str); nhi = atoi(star + 1); }
  if (nlo < 1 || nhi > nmax || nlo > nhi) throw InputError("Numeric index is out of bounds");
}

enum { MIX_GEOMETRIC = 0, MIX_ARITHMETIC = 1, MIX_SIXTHPOWER = 2 };

struct CoulTables {
  int nbits = 0, mask = 0, shift = 0;
  double tabinnersq = 0.0;
  std::vector<double> t[8];  // r, dr, f, df, c, dc, e, de
};

typedef union { int i; float f; } int_float_t;  // src/pair.h:208

class PairHost {
 public:
  polar_settings st;
  int ntypes = 0, allocated = 0;
  int mix_flag = MIX_GEOMETRIC, offset_flag = 0, tail_flag = 0;
  int ncoultablebits = 12;
  double tabinner = std::sqrt(2.0);
  double g_ewald = 0.0, qqrd2e = 0.0;
  double special_lj[4] = {1, 0, 0, 0}, special_coul[4] = {1, 0, 0, 0};
  std::vector<int> setflag;
  std::vector<double> epsilon, sigma, cut_lj, cut_ljsq, lj1, lj2, lj3, lj4, offset, cutsq;
  CoulTables tab;
  bool inited = false;

  PairHost() { settings_defaults(st); }
  int w() const { return ntypes + 1; }

  void allocate(int n) {  // PS.cpp:651-672
    ntypes = n;
    allocated = 1;
    size_t m = (size_t)(n + 1) * (n + 1);
    setflag.assign(m, 0);
    for (auto *v : {&epsilon, &sigma, &cut_lj, &cut_ljsq, &lj1, &lj2, &lj3, &lj4, &offset, &cutsq}) v->assign(m, 0.0);
  }

  // PS.cpp:678-766.  Note: the keyword scan starts at index 2, so keywords need both cutoffs.
  void settings(int narg, const char *const *arg) {
    if (narg < 1) throw InputError("Illegal pair_style command");
    st.cut_lj_global = numeric(arg[0]);
    if (narg == 1) st.cut_coul = st.cut_lj_global;
    else st.cut_coul = numeric(arg[1]);
    int iarg = 2;
    auto yesno = [&](const char *v) -> int {
      if (strcmp("yes", v) == 0) return 1;
      if (strcmp("no", v) == 0) return 0;
      throw InputError("Illegal pair_style command");
    };
    while (iarg < narg) {
      if (iarg + 2 > narg) throw InputError("Illegal pair_style command");
      const char *k = arg[iarg], *v = arg[iarg + 1];
      if (strcmp("precision", k) == 0) st.polar_precision = numeric(v);
      else if (strcmp("zodid", k) == 0) {
        if (st.polar_gs || st.polar_gs_ranked) throw InputError("Zodid doesn't work with polar_gs or polar_gs_ranked");
        st.zodid = yesno(v);
      } else if (strcmp("fixed_iteration", k) == 0) st.fixed_iteration = yesno(v);
      else if (strcmp("damp", k) == 0) st.polar_damp = numeric(v);
      else if (strcmp("max_iterations", k) == 0) st.iterations_max = inumeric(v);
      else if (strcmp("damp_type", k) == 0) {
        if (strcmp("exponential", v) == 0) st.damping_type = POLAR_DAMP_EXPONENTIAL;
        else if (strcmp("none", v) == 0) st.damping_type = POLAR_DAMP_NONE;
        else throw InputError("Illegal pair_style command");
      } else if (strcmp("polar_gs", k) == 0) {
        if (st.polar_gs_ranked) throw InputError("polar_gs and polar_gs_ranked are mutually exclusive");
        st.polar_gs = yesno(v);
      } else if (strcmp("polar_gs_ranked", k) == 0) {
        if (st.polar_gs) throw InputError("polar_gs and polar_gs_ranked are mutually exclusive");
        st.polar_gs_ranked = yesno(v);
      } else if (strcmp("polar_gamma", k) == 0) st.polar_gamma = numeric(v);
      else if (strcmp("debug", k) == 0) st.debug = yesno(v);
      else if (strcmp("use_previous", k) == 0) st.use_previous = yesno(v);
      else if (strcmp("dd_cutoff", k) == 0) st.dd_cutoff = numeric(v);  // extension keyword
      else if (strcmp("device_neigh", k) == 0) st.device_neigh = yesno(v);  // extension keyword
      else if (strcmp("restart_polar", k) == 0) st.restart_polar = yesno(v);  // extension keyword
      else if (strcmp("deterministic", k) == 0) st.deterministic = yesno(v) ? POLAR_DET_YES : POLAR_DET_NO;  // extension keyword (not given: POLAR_DET_AUTO)
      else if (strcmp("rccl_halo", k) == 0) st.rccl_halo = yesno(v);          // extension keyword
      else if (strcmp("polar_accel", k) == 0) {                               // extension keyword
        st.polar_accel = inumeric(v);
        if (st.polar_accel < 0 || st.polar_accel > POLAR_ACCEL_MAX) throw InputError("Illegal pair_style command");
      }
      else if (strcmp("polar_sor", k) == 0) {                                 // extension keyword
        st.polar_sor = numeric(v);
        if (!(st.polar_sor > 0.0 && st.polar_sor < 2.0)) throw InputError("Illegal pair_style command");
      }
      else throw InputError("Illegal pair_style command");
      iarg += 2;
    }
    if (allocated)  // PS.cpp:760-765: reset cutoffs that have been explicitly set
      for (int i = 1; i <= ntypes; i++)
        for (int j = i; j <= ntypes; j++)
          if (setflag[i * w() + j]) cut_lj[i * w() + j] = st.cut_lj_global;
    inited = false;
  }

  // PS.cpp:772-800
  void coeff(int n, int narg, const char *const *arg) {
    if (narg < 4 || narg > 5) throw InputError("Incorrect args for pair coefficients");
    if (!allocated) allocate(n);
    int ilo, ihi, jlo, jhi;
    bounds(arg[0], ntypes, ilo, ihi);
    bounds(arg[1], ntypes, jlo, jhi);
    double epsilon_one = numeric(arg[2]);
    double sigma_one = numeric(arg[3]);
    double cut_lj_one = st.cut_lj_global;
    if (narg == 5) cut_lj_one = numeric(arg[4]);
    int count = 0;
    for (int i = ilo; i <= ihi; i++)
      for (int j = (jlo > i ? jlo : i); j <= jhi; j++) {
        epsilon[i * w() + j] = epsilon_one;
        sigma[i * w() + j] = sigma_one;
        cut_lj[i * w() + j] = cut_lj_one;
        setflag[i * w() + j] = 1;
        count++;
      }
    if (count == 0) throw InputError("Incorrect args for pair coefficients");
    inited = false;
  }

  // src/pair.cpp:125-185 (keywords that matter to this style)
  void modify(int narg, const char *const *arg) {
    if (narg == 0) throw InputError("Illegal pair_modify command");
    int iarg = 0;
    while (iarg < narg) {
      if (iarg + 2 > narg) throw InputError("Illegal pair_modify command");
      const char *k = arg[iarg], *v = arg[iarg + 1];
      if (strcmp(k, "mix") == 0) {
        if (strcmp(v, "geometric") == 0) mix_flag = MIX_GEOMETRIC;
        else if (strcmp(v, "arithmetic") == 0) mix_flag = MIX_ARITHMETIC;
        else if (strcmp(v, "sixthpower") == 0) mix_flag = MIX_SIXTHPOWER;
        else throw InputError("Illegal pair_modify command");
      } else if (strcmp(k, "shift") == 0) {
        if (strcmp(v, "yes") == 0) offset_flag = 1;
        else if (strcmp(v, "no") == 0) offset_flag = 0;
        else throw InputError("Illegal pair_modify command");
      } else if (strcmp(k, "table") == 0) {
        ncoultablebits = inumeric(v);
        if (ncoultablebits > (int)(sizeof(float) * CHAR_BIT)) throw InputError("Too many total bits for bitmapped lookup table");
      } else if (strcmp(k, "tabinner") == 0) {
        tabinner = numeric(v);
      } else if (strcmp(k, "tail") == 0) {
        if (strcmp(v, "yes") == 0) tail_flag = 1;
        else if (strcmp(v, "no") == 0) tail_flag = 0;
        else throw InputError("Illegal pair_modify command");
      } else throw InputError("Illegal pair_modify command");
      iarg += 2;
    }
    inited = false;
  }

  double mix_energy(double e1, double e2, double s1, double s2) const {  // src/pair.cpp:660-671
    if (mix_flag == MIX_GEOMETRIC || mix_flag == MIX_ARITHMETIC) return std::sqrt(e1 * e2);
    return 2.0 * std::sqrt(e1 * e2) * std::pow(s1, 3.0) * std::pow(s2, 3.0) / (std::pow(s1, 6.0) + std::pow(s2, 6.0));
  }
  double mix_distance(double s1, double s2) const {  // src/pair.cpp:677-686
    if (mix_flag == MIX_GEOMETRIC) return std::sqrt(s1 * s2);
    if (mix_flag == MIX_ARITHMETIC) return 0.5 * (s1 + s2);
    return std::pow(0.5 * (std::pow(s1, 6.0) + std::pow(s2, 6.0)), 1.0 / 6.0);
  }

  double init_one(int i, int j) {  // PS.cpp:858-921 (tail correction sums are a caller-side quantity)
    int ij = i * w() + j, ji = j * w() + i;
    if (setflag[ij] == 0) {
      epsilon[ij] = mix_energy(epsilon[i * w() + i], epsilon[j * w() + j], sigma[i * w() + i], sigma[j * w() + j]);
      sigma[ij] = mix_distance(sigma[i * w() + i], sigma[j * w() + j]);
      cut_lj[ij] = mix_distance(cut_lj[i * w() + i], cut_lj[j * w() + j]);
    }
    double cut = cut_lj[ij] > st.cut_coul ? cut_lj[ij] : st.cut_coul;  // qdist = 0
    cut_ljsq[ij] = cut_lj[ij] * cut_lj[ij];
    lj1[ij] = 48.0 * epsilon[ij] * std::pow(sigma[ij], 12.0);
    lj2[ij] = 24.0 * epsilon[ij] * std::pow(sigma[ij], 6.0);
    lj3[ij] = 4.0 * epsilon[ij] * std::pow(sigma[ij], 12.0);
    lj4[ij] = 4.0 * epsilon[ij] * std::pow(sigma[ij], 6.0);
    if (offset_flag && cut_lj[ij] > 0.0) {
      double ratio = sigma[ij] / cut_lj[ij];
      offset[ij] = 4.0 * epsilon[ij] * (std::pow(ratio, 12.0) - std::pow(ratio, 6.0));
    } else offset[ij] = 0.0;
    cut_ljsq[ji] = cut_ljsq[ij]; lj1[ji] = lj1[ij]; lj2[ji] = lj2[ij];
    lj3[ji] = lj3[ij]; lj4[ji] = lj4[ij]; offset[ji] = offset[ij];
    return cut;
  }

  void init(double g, double qq, const double *slj, const double *scoul) {  // src/pair.cpp:189-263
    if (offset_flag && tail_flag) throw InputError("Cannot have both pair_modify shift and tail set to yes");
    if (!allocated) throw InputError("All pair coeffs are not set");
    for (int i = 1; i <= ntypes; i++)
      if (setflag[i * w() + i] == 0) throw InputError("All pair coeffs are not set");
    g_ewald = g; qqrd2e = qq;
    for (int k = 0; k < 4; k++) { special_lj[k] = slj[k]; special_coul[k] = scoul[k]; }
    tab = CoulTables();  // PS.cpp:851 builds the tables here; this library receives them through polar_set_coul
    for (int i = 1; i <= ntypes; i++)
      for (int j = i; j <= ntypes; j++) {
        double cut = init_one(i, j);
        cutsq[i * w() + j] = cutsq[j * w() + i] = cut * cut;
      }
    inited = true;
  }

  // The Coulomb lookup tables (Pair::init_tables, src/pair.cpp:313-520) are NOT generated here: SURVEY 8(b) keeps them
  // LAMMPS host code.  The shim hands over the arrays Pair::init_tables built (polar_set_coul); tests and the bench
  // hand over workload.init_coul_tables' (test infrastructure).  This class only keeps a copy for single().
  void set_tables(int nbits, int mask, int shift, double tabinnersq, const double *const t[8]) {
    tab = CoulTables();
    if (nbits <= 0) return;
    if (nbits > 24) throw InputError("Too many bits for lookup table");
    const size_t ntable = (size_t)1 << nbits;
    for (int k = 0; k < 8; k++) {
      if (!t[k]) throw InputError("polar_set_coul: null Coulomb table");
      tab.t[k].assign(t[k], t[k] + ntable);
    }
    tab.nbits = nbits; tab.mask = mask; tab.shift = shift; tab.tabinnersq = tabinnersq;
  }

  // PS.cpp:1035-1097
  double single(double qi, double qj, int itype, int jtype, double rsq, double factor_coul, double factor_lj,
                double &fforce) const {
    const double EWALD_F = 1.12837917, EWALD_P = 0.3275911, A1 = 0.254829592, A2 = -0.284496736, A3 = 1.421413741,
                 A4 = -1.453152027, A5 = 1.061405429;
    double r2inv = 1.0 / rsq, forcecoul, forcelj, prefactor = 0, erfc_ = 0, fraction = 0, r6inv = 0;
    int itable = 0;
    const double cut_coulsq = st.cut_coul * st.cut_coul;
    const int ij = itype * w() + jtype;
    if (rsq < cut_coulsq) {
      if (ncoultablebits && !tab.nbits) throw InputError("Coulomb tables were not handed over (polar_set_coul after polar_pair_init)");
      if (!tab.nbits || rsq <= tab.tabinnersq) {
        double r = std::sqrt(rsq), grij = g_ewald * r, expm2 = std::exp(-grij * grij);
        double t = 1.0 / (1.0 + EWALD_P * grij);
        erfc_ = t * (A1 + t * (A2 + t * (A3 + t * (A4 + t * A5)))) * expm2;
        prefactor = qqrd2e * qi * qj / r;
        forcecoul = prefactor * (erfc_ + EWALD_F * grij * expm2);
        if (factor_coul < 1.0) forcecoul -= (1.0 - factor_coul) * prefactor;
      } else {
        int_float_t u;
        u.f = (float)rsq;
        itable = (u.i & tab.mask) >> tab.shift;
        fraction = (u.f - tab.t[0][itable]) * tab.t[1][itable];
        double table = tab.t[2][itable] + fraction * tab.t[3][itable];
        forcecoul = qi * qj * table;
        if (factor_coul < 1.0) {
          table = tab.t[4][itable] + fraction * tab.t[5][itable];
          prefactor = qi * qj * table;
          forcecoul -= (1.0 - factor_coul) * prefactor;
        }
      }
    } else forcecoul = 0.0;
    if (rsq < cut_ljsq[ij]) {
      r6inv = r2inv * r2inv * r2inv;
      forcelj = r6inv * (lj1[ij] * r6inv - lj2[ij]);
    } else forcelj = 0.0;
    fforce = (forcecoul + factor_lj * forcelj) * r2inv;
    double eng = 0.0;
    if (rsq < cut_coulsq) {
      double phicoul;
      if (!tab.nbits || rsq <= tab.tabinnersq) phicoul = prefactor * erfc_;
      else phicoul = qi * qj * (tab.t[6][itable] + fraction * tab.t[7][itable]);
      if (factor_coul < 1.0) phicoul -= (1.0 - factor_coul) * prefactor;
      eng += phicoul;
    }
    if (rsq < cut_ljsq[ij]) eng += factor_lj * (r6inv * (lj3[ij] * r6inv - lj4[ij]) - offset[ij]);
    return eng;
  }
};

}  // namespace polar
