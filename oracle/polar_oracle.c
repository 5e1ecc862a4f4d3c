/* polar_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see polar_oracle.h).
 *
 * CPU restatement of the reference lj/cut/coul/long/polarization pair style.
 * "PS.cpp" = /root/reference/src/pair_lj_cut_coul_long_polarization.cpp.
 * Arithmetic order follows the reference so results agree to rounding.
 */
#define _POSIX_C_SOURCE 200809L
#include "polar_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define EWALD_F 1.12837917 /* PS.cpp:43-49 */
#define EWALD_P 0.3275911
#define A1 0.254829592
#define A2 -0.284496736
#define A3 1.421413741
#define A4 -1.453152027
#define A5 1.061405429
#define MY_ISPI4 1.12837916709551257390 /* math_const.h: 1/sqrt(pi/4) */

#define SBBITS 30              /* lmptype.h:58 */
#define NEIGHMASK 0x3FFFFFFF   /* lmptype.h:59 */

typedef union { int i; float f; } int_float_t; /* pair.h:208 */

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------ a9 --
 * Domain::closest_image(xi,xj,xjimage)  domain.cpp:1220-1312
 * The add/subtract sequence is kept (it is not a plain nearbyint wrap).   */
void orc_closest_image(const orc_system *s, const double *xi, const double *xj, double *xjimage) {
  double dx = xj[0] - xi[0];
  double dy = xj[1] - xi[1];
  double dz = xj[2] - xi[2];
  const double xprd = s->prd[0], yprd = s->prd[1], zprd = s->prd[2];
  const double xh = 0.5 * xprd, yh = 0.5 * yprd, zh = 0.5 * zprd;

  if (!s->triclinic) {
    if (s->periodic[0]) {
      if (dx < 0.0) { while (dx < 0.0) dx += xprd; if (dx > xh) dx -= xprd; }
      else          { while (dx > 0.0) dx -= xprd; if (dx < -xh) dx += xprd; }
    }
    if (s->periodic[1]) {
      if (dy < 0.0) { while (dy < 0.0) dy += yprd; if (dy > yh) dy -= yprd; }
      else          { while (dy > 0.0) dy -= yprd; if (dy < -yh) dy += yprd; }
    }
    if (s->periodic[2]) {
      if (dz < 0.0) { while (dz < 0.0) dz += zprd; if (dz > zh) dz -= zprd; }
      else          { while (dz > 0.0) dz -= zprd; if (dz < -zh) dz += zprd; }
    }
  } else {
    const double xy = s->tilt[0], xz = s->tilt[1], yz = s->tilt[2];
    if (s->periodic[2]) {
      if (dz < 0.0) {
        while (dz < 0.0) { dz += zprd; dy += yz; dx += xz; }
        if (dz > zh) { dz -= zprd; dy -= yz; dx -= xz; }
      } else {
        while (dz > 0.0) { dz -= zprd; dy -= yz; dx -= xz; }
        if (dz < -zh) { dz += zprd; dy += yz; dx += xz; }
      }
    }
    if (s->periodic[1]) {
      if (dy < 0.0) {
        while (dy < 0.0) { dy += yprd; dx += xy; }
        if (dy > yh) { dy -= yprd; dx -= xy; }
      } else {
        while (dy > 0.0) { dy -= yprd; dx -= xy; }
        if (dy < -yh) { dy += yprd; dx += xy; }
      }
    }
    if (s->periodic[0]) {
      if (dx < 0.0) { while (dx < 0.0) dx += xprd; if (dx > xh) dx -= xprd; }
      else          { while (dx > 0.0) dx -= xprd; if (dx < -xh) dx += xprd; }
    }
  }
  xjimage[0] = xi[0] + dx;
  xjimage[1] = xi[1] + dy;
  xjimage[2] = xi[2] + dz;
}

/* ------------------------------------------------------------------------
 * Pair::init_bitmap pair.cpp:1676-1723 + Pair::init_tables pair.cpp:313-520
 * (branch cut_respa == NULL, msmflag == 0).                               */
static void init_bitmap(double inner, double outer, int ntablebits, int *masklo, int *maskhi,
                        int *nmask, int *nshiftbits) {
  int nlowermin = 1;
  while (!((pow(2.0, (double)nlowermin) <= inner * inner) &&
           (pow(2.0, (double)nlowermin + 1.0) > inner * inner))) {
    if (pow(2.0, (double)nlowermin) <= inner * inner) nlowermin++;
    else nlowermin--;
  }
  int nexpbits = 0;
  double required_range = outer * outer / pow(2.0, (double)nlowermin);
  double available_range = 2.0;
  while (available_range < required_range) {
    nexpbits++;
    available_range = pow(2.0, pow(2.0, (double)nexpbits));
  }
  int nmantbits = ntablebits - nexpbits;
  *nshiftbits = FLT_MANT_DIG - (nmantbits + 1);
  int m = 1;
  for (int j = 0; j < ntablebits + *nshiftbits; j++) m *= 2;
  m -= 1;
  *nmask = m;
  int_float_t u;
  u.f = (float)(outer * outer);
  *maskhi = u.i & ~m;
  u.f = (float)(inner * inner);
  *masklo = u.i & ~m;
}

int orc_init_tables(double cut_coul, double g_ewald, double qqrd2e, double tabinner,
                    int ncoultablebits, int *ncoulmask, int *ncoulshiftbits, double *tabinnersq_out,
                    double *tables) {
  int masklo, maskhi, nmask, nshift;
  double cut_coulsq = cut_coul * cut_coul;
  double tabinnersq = tabinner * tabinner;
  init_bitmap(tabinner, cut_coul, ncoultablebits, &masklo, &maskhi, &nmask, &nshift);
  int ntable = 1;
  for (int i = 0; i < ncoultablebits; i++) ntable *= 2;
  double *rtable = tables, *drtable = tables + ntable, *ftable = tables + 2 * ntable,
         *dftable = tables + 3 * ntable, *ctable = tables + 4 * ntable,
         *dctable = tables + 5 * ntable, *etable = tables + 6 * ntable,
         *detable = tables + 7 * ntable;

  int_float_t rsq_lookup, minrsq_lookup;
  minrsq_lookup.i = 0 << nshift;
  minrsq_lookup.i |= maskhi;
  for (int i = 0; i < ntable; i++) {
    rsq_lookup.i = i << nshift;
    rsq_lookup.i |= masklo;
    if (rsq_lookup.f < tabinnersq) {
      rsq_lookup.i = i << nshift;
      rsq_lookup.i |= maskhi;
    }
    double r = sqrtf(rsq_lookup.f);
    double grij = g_ewald * r;
    double expm2 = exp(-grij * grij);
    double derfc = erfc(grij);
    rtable[i] = rsq_lookup.f;
    ctable[i] = qqrd2e / r;
    ftable[i] = qqrd2e / r * (derfc + MY_ISPI4 * grij * expm2);
    etable[i] = qqrd2e / r * derfc;
    if (rsq_lookup.f < minrsq_lookup.f) minrsq_lookup.f = rsq_lookup.f;
  }
  tabinnersq = minrsq_lookup.f;
  int ntablem1 = ntable - 1;
  for (int i = 0; i < ntablem1; i++) {
    drtable[i] = 1.0 / (rtable[i + 1] - rtable[i]);
    dftable[i] = ftable[i + 1] - ftable[i];
    dctable[i] = ctable[i + 1] - ctable[i];
    detable[i] = etable[i + 1] - etable[i];
  }
  drtable[ntablem1] = 1.0 / (rtable[0] - rtable[ntablem1]);
  dftable[ntablem1] = ftable[0] - ftable[ntablem1];
  dctable[ntablem1] = ctable[0] - ctable[ntablem1];
  detable[ntablem1] = etable[0] - etable[ntablem1];

  int itablemin = minrsq_lookup.i & nmask;
  itablemin >>= nshift;
  int itablemax = itablemin - 1;
  if (itablemin == 0) itablemax = ntablem1;
  rsq_lookup.i = itablemax << nshift;
  rsq_lookup.i |= maskhi;
  if (rsq_lookup.f < cut_coulsq) {
    rsq_lookup.f = (float)cut_coulsq;
    double r = sqrtf(rsq_lookup.f);
    double grij = g_ewald * r;
    double expm2 = exp(-grij * grij);
    double derfc = erfc(grij);
    double c_tmp = qqrd2e / r;
    double f_tmp = qqrd2e / r * (derfc + MY_ISPI4 * grij * expm2);
    double e_tmp = qqrd2e / r * derfc;
    drtable[itablemax] = 1.0 / (rsq_lookup.f - rtable[itablemax]);
    dftable[itablemax] = f_tmp - ftable[itablemax];
    dctable[itablemax] = c_tmp - ctable[itablemax];
    detable[itablemax] = e_tmp - etable[itablemax];
  }
  *ncoulmask = nmask;
  *ncoulshiftbits = nshift;
  *tabinnersq_out = tabinnersq;
  return ntable;
}

/* ------------------------------------------------------------------------
 * init_one for every type pair: PS.cpp:858-921, mixing pair.cpp:660-690,
 * cutsq = init_one()^2 as Pair::init does (pair.cpp:230-260).             */
static double mix_energy(int mix, double e1, double e2, double s1, double s2) {
  if (mix == 0 || mix == 1) return sqrt(e1 * e2);
  return 2.0 * sqrt(e1 * e2) * pow(s1, 3.0) * pow(s2, 3.0) / (pow(s1, 6.0) + pow(s2, 6.0));
}
static double mix_distance(int mix, double s1, double s2) {
  if (mix == 0) return sqrt(s1 * s2);
  if (mix == 1) return 0.5 * (s1 + s2);
  return pow(0.5 * (pow(s1, 6.0) + pow(s2, 6.0)), 1.0 / 6.0);
}

void orc_init_one_all(int n, const int *setflag, double *eps, double *sig, double *cutlj,
                      int mix_flag, int offset_flag, double cut_coul, double *lj1, double *lj2,
                      double *lj3, double *lj4, double *offset, double *cut_ljsq, double *cutsq) {
  int w = n + 1;
  for (int i = 1; i <= n; i++)
    for (int j = i; j <= n; j++) {
      int ij = i * w + j, ji = j * w + i;
      if (!setflag[ij]) {
        eps[ij] = mix_energy(mix_flag, eps[i * w + i], eps[j * w + j], sig[i * w + i], sig[j * w + j]);
        sig[ij] = mix_distance(mix_flag, sig[i * w + i], sig[j * w + j]);
        cutlj[ij] = mix_distance(mix_flag, cutlj[i * w + i], cutlj[j * w + j]);
      }
      double cut = cutlj[ij] > cut_coul ? cutlj[ij] : cut_coul; /* qdist = 0 */
      cut_ljsq[ij] = cutlj[ij] * cutlj[ij];
      lj1[ij] = 48.0 * eps[ij] * pow(sig[ij], 12.0);
      lj2[ij] = 24.0 * eps[ij] * pow(sig[ij], 6.0);
      lj3[ij] = 4.0 * eps[ij] * pow(sig[ij], 12.0);
      lj4[ij] = 4.0 * eps[ij] * pow(sig[ij], 6.0);
      if (offset_flag && cutlj[ij] > 0.0) {
        double ratio = sig[ij] / cutlj[ij];
        offset[ij] = 4.0 * eps[ij] * (pow(ratio, 12.0) - pow(ratio, 6.0));
      } else offset[ij] = 0.0;
      cutsq[ij] = cut * cut;
      cut_ljsq[ji] = cut_ljsq[ij]; lj1[ji] = lj1[ij]; lj2[ji] = lj2[ij];
      lj3[ji] = lj3[ij]; lj4[ji] = lj4[ij]; offset[ji] = offset[ij]; cutsq[ji] = cutsq[ij];
    }
}

/* ------------------------------------------------------------------------
 * Extension support: minimum-image cell list -> CSR full neighbor list over
 * LOCAL atoms (used only when dd_cutoff > 0).  Orthogonal periodic boxes.   */
typedef struct {
  long long *first; /* [n+1] */
  int *j;           /* neighbor local index */
  double *d;        /* [3] per pair: x_i - x_j(image), closest_image arithmetic */
  double *rsq;
  long long npairs;
} nbr_list;

static void nbr_free(nbr_list *L) {
  free(L->first); free(L->j); free(L->d); free(L->rsq);
  memset(L, 0, sizeof(*L));
}

static void nbr_build_allpairs(const orc_system *s, double cut, nbr_list *L);

static void nbr_build(const orc_system *s, double cut, nbr_list *L) {
  const int n = s->nlocal;
  const double *x = s->x;
  if (s->triclinic) { nbr_build_allpairs(s, cut, L); return; } /* the grid below is orthogonal */
  int nc[3];
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) {
      if (x[3 * i + k] < lo[k]) lo[k] = x[3 * i + k];
      if (x[3 * i + k] > hi[k]) hi[k] = x[3 * i + k];
    }
  double span[3];
  for (int k = 0; k < 3; k++) {
    span[k] = s->periodic[k] ? s->prd[k] : (hi[k] - lo[k]) + 1e-9;
    nc[k] = (int)floor(span[k] / cut);
    if (nc[k] < 1) nc[k] = 1;
  }
  long long ncell = (long long)nc[0] * nc[1] * nc[2];
  int *cell_of = (int *)malloc(sizeof(int) * (size_t)n);
  int *cnt = (int *)calloc((size_t)ncell + 1, sizeof(int));
  for (int i = 0; i < n; i++) {
    int c[3];
    for (int k = 0; k < 3; k++) {
      double fr = (x[3 * i + k] - lo[k]) / span[k];
      fr -= floor(fr);
      c[k] = (int)(fr * nc[k]);
      if (c[k] >= nc[k]) c[k] = nc[k] - 1;
    }
    cell_of[i] = (c[2] * nc[1] + c[1]) * nc[0] + c[0];
    cnt[cell_of[i] + 1]++;
  }
  for (long long c = 0; c < ncell; c++) cnt[c + 1] += cnt[c];
  int *order = (int *)malloc(sizeof(int) * (size_t)n);
  int *fill = (int *)calloc((size_t)ncell, sizeof(int));
  for (int i = 0; i < n; i++) order[cnt[cell_of[i]] + fill[cell_of[i]]++] = i;
  free(fill);

  const double cutsq = cut * cut;
  L->first = (long long *)malloc(sizeof(long long) * ((size_t)n + 1));
  /* two passes: count, then fill */
  for (int pass = 0; pass < 2; pass++) {
    long long tot = 0;
    for (int i = 0; i < n; i++) {
      if (pass == 1) tot = L->first[i];
      int ci = cell_of[i];
      int c0 = ci % nc[0], c1 = (ci / nc[0]) % nc[1], c2 = ci / (nc[0] * nc[1]);
      /* visited-cell dedup for tiny grids */
      int seen[27], nseen = 0;
      for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
          for (int dx = -1; dx <= 1; dx++) {
            int b0 = c0 + dx, b1 = c1 + dy, b2 = c2 + dz;
            if (s->periodic[0]) b0 = (b0 + nc[0]) % nc[0]; else if (b0 < 0 || b0 >= nc[0]) continue;
            if (s->periodic[1]) b1 = (b1 + nc[1]) % nc[1]; else if (b1 < 0 || b1 >= nc[1]) continue;
            if (s->periodic[2]) b2 = (b2 + nc[2]) % nc[2]; else if (b2 < 0 || b2 >= nc[2]) continue;
            int cj = (b2 * nc[1] + b1) * nc[0] + b0;
            int dup = 0;
            for (int t = 0; t < nseen; t++) if (seen[t] == cj) dup = 1;
            if (dup) continue;
            seen[nseen++] = cj;
            for (int t = cnt[cj]; t < cnt[cj + 1]; t++) {
              int j = order[t];
              if (j == i) continue;
              double xim[3];
              orc_closest_image(s, &x[3 * i], &x[3 * j], xim);
              double ddx = x[3 * i] - xim[0], ddy = x[3 * i + 1] - xim[1], ddz = x[3 * i + 2] - xim[2];
              double rsq = ddx * ddx + ddy * ddy + ddz * ddz;
              if (rsq <= cutsq) {
                if (pass == 1) {
                  L->j[tot] = j; L->rsq[tot] = rsq;
                  L->d[3 * tot] = ddx; L->d[3 * tot + 1] = ddy; L->d[3 * tot + 2] = ddz;
                }
                tot++;
              }
            }
          }
      if (pass == 0) L->first[i + 1] = tot;
    }
    if (pass == 0) {
      L->first[0] = 0;
      L->npairs = L->first[n];
      L->j = (int *)malloc(sizeof(int) * (size_t)(L->npairs + 1));
      L->d = (double *)malloc(sizeof(double) * 3 * (size_t)(L->npairs + 1));
      L->rsq = (double *)malloc(sizeof(double) * (size_t)(L->npairs + 1));
    }
  }
  /* sort each row by j so that accumulation order is index order like the reference loops */
  for (int i = 0; i < n; i++) {
    long long a = L->first[i], b = L->first[i + 1];
    for (long long p = a + 1; p < b; p++) { /* insertion sort: rows arrive nearly cell-ordered */
      int jj = L->j[p]; double rr = L->rsq[p];
      double d0 = L->d[3 * p], d1 = L->d[3 * p + 1], d2 = L->d[3 * p + 2];
      long long q = p - 1;
      while (q >= a && L->j[q] > jj) {
        L->j[q + 1] = L->j[q]; L->rsq[q + 1] = L->rsq[q];
        L->d[3 * q + 3] = L->d[3 * q]; L->d[3 * q + 4] = L->d[3 * q + 1]; L->d[3 * q + 5] = L->d[3 * q + 2];
        q--;
      }
      L->j[q + 1] = jj; L->rsq[q + 1] = rr;
      L->d[3 * q + 3] = d0; L->d[3 * q + 4] = d1; L->d[3 * q + 5] = d2;
    }
  }
  free(cell_of); free(cnt); free(order);
}

/* ------------------------------------------------------------------ a2 --
 * PS.cpp:192-227: rmin over raw (non-minimum-image) distances to locals AND ghosts, then
 * rank_metric[i] = sum alpha_i*alpha_j over j within 1.5*rmin (different molecule or mol 0). */
void orc_rank_metric(const orc_system *s, double *rank_metric, double *rmin_out) {
  const int nlocal = s->nlocal, ntotal = s->nlocal + s->nghost;
  const double *x = s->x, *a = s->alpha;
  const int *mol = s->molecule;
  double rmin = 1000.0;
  if (s->dd_cutoff > 0.0) {
    /* extension: same definition over minimum-image pairs from the cell list */
    nbr_list L; memset(&L, 0, sizeof(L));
    double cut = s->dd_cutoff > s->cut_coul ? s->dd_cutoff : s->cut_coul;
    nbr_build(s, cut, &L);
    for (int i = 0; i < nlocal; i++)
      for (long long p = L.first[i]; p < L.first[i + 1]; p++) {
        int j = L.j[p];
        double r = sqrt(L.rsq[p]);
        if (a[i] > 0 && a[j] > 0 && rmin > r && ((mol[i] != mol[j]) || mol[i] == 0)) rmin = r;
      }
    for (int i = 0; i < nlocal; i++) {
      rank_metric[i] = 0;
      for (long long p = L.first[i]; p < L.first[i + 1]; p++) {
        int j = L.j[p];
        double r = sqrt(L.rsq[p]);
        if (rmin * 1.5 > r && ((mol[i] != mol[j]) || mol[i] == 0)) rank_metric[i] += a[i] * a[j];
      }
    }
    nbr_free(&L);
    *rmin_out = rmin;
    return;
  }
  for (int i = 0; i < nlocal; i++)
    for (int j = 0; j < ntotal; j++)
      if (i != j) {
        double r = sqrt(pow(x[3 * i] - x[3 * j], 2) + pow(x[3 * i + 1] - x[3 * j + 1], 2) +
                        pow(x[3 * i + 2] - x[3 * j + 2], 2));
        if (a[i] > 0 && a[j] > 0 && rmin > r && ((mol[i] != mol[j]) || mol[i] == 0)) rmin = r;
      }
  for (int i = 0; i < nlocal; i++) rank_metric[i] = 0;
  for (int i = 0; i < nlocal; i++)
    for (int j = 0; j < ntotal; j++)
      if (i != j) {
        double r = sqrt(pow(x[3 * i] - x[3 * j], 2) + pow(x[3 * i + 1] - x[3 * j + 1], 2) +
                        pow(x[3 * i + 2] - x[3 * j + 2], 2));
        if (rmin * 1.5 > r && ((mol[i] != mol[j]) || mol[i] == 0)) rank_metric[i] += a[i] * a[j];
      }
  *rmin_out = rmin;
}

/* ------------------------------------------------------------------ a3 --
 * PS.cpp:232-321: stock lj/cut/coul/long half-list loop.                   */
void orc_ljcoul(const orc_system *s, int eflag, int vflag_pairwise, double *f, double *eng_vdwl,
                double *eng_coul, double *virial, double *eatom, double *vatom) {
  const int w = s->ntypes + 1, nlocal = s->nlocal;
  const double *x = s->x, *q = s->q;
  const double cut_coulsq = s->cut_coul * s->cut_coul;
  const double qqrd2e = s->qqrd2e, g_ewald = s->g_ewald;
  double evdwl = 0.0, ecoul = 0.0;
  for (int ii = 0; ii < s->inum; ii++) {
    int i = s->ilist[ii];
    double qtmp = q[i], xtmp = x[3 * i], ytmp = x[3 * i + 1], ztmp = x[3 * i + 2];
    int itype = s->type[i];
    const int *jlist = s->neigh + s->firstneigh[i];
    int jnum = s->numneigh[i];
    for (int jj = 0; jj < jnum; jj++) {
      int j = jlist[jj];
      double factor_lj = s->special_lj[(j >> SBBITS) & 3];
      double factor_coul = s->special_coul[(j >> SBBITS) & 3];
      j &= NEIGHMASK;
      double delx = xtmp - x[3 * j], dely = ytmp - x[3 * j + 1], delz = ztmp - x[3 * j + 2];
      double rsq = delx * delx + dely * dely + delz * delz;
      int jtype = s->type[j];
      if (rsq < s->cutsq[itype * w + jtype]) {
        double r2inv = 1.0 / rsq, forcecoul, forcelj, prefactor = 0.0, erfc_ = 0.0, fraction = 0.0;
        double r6inv = 0.0;
        int itable = 0;
        if (rsq < cut_coulsq) {
          if (!s->ncoultablebits || rsq <= s->tabinnersq) {
            double r = sqrt(rsq);
            double grij = g_ewald * r;
            double expm2 = exp(-grij * grij);
            double t = 1.0 / (1.0 + EWALD_P * grij);
            erfc_ = t * (A1 + t * (A2 + t * (A3 + t * (A4 + t * A5)))) * expm2;
            prefactor = qqrd2e * qtmp * q[j] / r;
            forcecoul = prefactor * (erfc_ + EWALD_F * grij * expm2);
            if (factor_coul < 1.0) forcecoul -= (1.0 - factor_coul) * prefactor;
          } else {
            int_float_t rsq_lookup;
            rsq_lookup.f = (float)rsq;
            itable = rsq_lookup.i & s->ncoulmask;
            itable >>= s->ncoulshiftbits;
            fraction = (rsq_lookup.f - s->rtable[itable]) * s->drtable[itable];
            double table = s->ftable[itable] + fraction * s->dftable[itable];
            forcecoul = qtmp * q[j] * table;
            if (factor_coul < 1.0) {
              table = s->ctable[itable] + fraction * s->dctable[itable];
              prefactor = qtmp * q[j] * table;
              forcecoul -= (1.0 - factor_coul) * prefactor;
            }
          }
        } else forcecoul = 0.0;

        if (rsq < s->cut_ljsq[itype * w + jtype]) {
          r6inv = r2inv * r2inv * r2inv;
          forcelj = r6inv * (s->lj1[itype * w + jtype] * r6inv - s->lj2[itype * w + jtype]);
        } else forcelj = 0.0;

        double fpair = (forcecoul + factor_lj * forcelj) * r2inv;
        f[3 * i] += delx * fpair; f[3 * i + 1] += dely * fpair; f[3 * i + 2] += delz * fpair;
        if (s->newton_pair || j < nlocal) {
          f[3 * j] -= delx * fpair; f[3 * j + 1] -= dely * fpair; f[3 * j + 2] -= delz * fpair;
        }
        if (eflag) {
          if (rsq < cut_coulsq) {
            if (!s->ncoultablebits || rsq <= s->tabinnersq) ecoul = prefactor * erfc_;
            else {
              double table = s->etable[itable] + fraction * s->detable[itable];
              ecoul = qtmp * q[j] * table;
            }
            if (factor_coul < 1.0) ecoul -= (1.0 - factor_coul) * prefactor;
          } else ecoul = 0.0;
          if (rsq < s->cut_ljsq[itype * w + jtype]) {
            evdwl = r6inv * (s->lj3[itype * w + jtype] * r6inv - s->lj4[itype * w + jtype]) -
                    s->offset[itype * w + jtype];
            evdwl *= factor_lj;
          } else evdwl = 0.0;
        }
        /* ev_tally pair.cpp:854-950, global parts only */
        if (eflag) {
          if (s->newton_pair) { *eng_vdwl += evdwl; *eng_coul += ecoul; }
          else {
            if (i < nlocal) { *eng_vdwl += 0.5 * evdwl; *eng_coul += 0.5 * ecoul; }
            if (j < nlocal) { *eng_vdwl += 0.5 * evdwl; *eng_coul += 0.5 * ecoul; }
          }
        }
        if (vflag_pairwise) {
          double v[6] = {delx * delx * fpair, dely * dely * fpair, delz * delz * fpair,
                         delx * dely * fpair, delx * delz * fpair, dely * delz * fpair};
          if (s->newton_pair) for (int k = 0; k < 6; k++) virial[k] += v[k];
          else {
            if (i < nlocal) for (int k = 0; k < 6; k++) virial[k] += 0.5 * v[k];
            if (j < nlocal) for (int k = 0; k < 6; k++) virial[k] += 0.5 * v[k];
          }
        }
        /* per-atom parts of ev_tally, pair.cpp:881-885 (eatom) and 925-942 (vatom) */
        if (eatom) {
          double epairhalf = 0.5 * (evdwl + ecoul);
          if (s->newton_pair || i < nlocal) eatom[i] += epairhalf;
          if (s->newton_pair || j < nlocal) eatom[j] += epairhalf;
        }
        if (vatom) {
          double v[6] = {delx * delx * fpair, dely * dely * fpair, delz * delz * fpair,
                         delx * dely * fpair, delx * delz * fpair, dely * delz * fpair};
          if (s->newton_pair || i < nlocal) for (int k = 0; k < 6; k++) vatom[6 * i + k] += 0.5 * v[k];
          if (s->newton_pair || j < nlocal) for (int k = 0; k < 6; k++) vatom[6 * j + k] += 0.5 * v[k];
        }
      }
    }
  }
}

/* ------------------------------------------------------------------ a4 --
 * PS.cpp:324-361: shifted-force ("wolf, no damping") static field, i<j minimum image,
 * rsq <= cut_coulsq (note <=), molecule exclusion unless molecule id 0.     */
void orc_static_field(const orc_system *s, double *ef_static) {
  const int nlocal = s->nlocal;
  const double *x = s->x, *q = s->q;
  const int *mol = s->molecule;
  const double cut_coulsq = s->cut_coul * s->cut_coul;
  const double f_shift = -1.0 / (s->cut_coul * s->cut_coul);
  for (int i = 0; i < 3 * nlocal; i++) ef_static[i] = 0; /* PS.cpp:151-156 */
  if (s->dd_cutoff > 0.0) {
    nbr_list L; memset(&L, 0, sizeof(L));
    nbr_build(s, s->cut_coul, &L);
    for (int i = 0; i < nlocal; i++)
      for (long long p = L.first[i]; p < L.first[i + 1]; p++) {
        int j = L.j[p];
        double rsq = L.rsq[p];
        if (rsq <= cut_coulsq && ((mol[i] != mol[j]) || mol[i] == 0)) {
          double r = sqrt(rsq);
          double ef_temp = (1.0 / rsq + f_shift) * 1.0 / r;
          ef_static[3 * i] += ef_temp * q[j] * L.d[3 * p];
          ef_static[3 * i + 1] += ef_temp * q[j] * L.d[3 * p + 1];
          ef_static[3 * i + 2] += ef_temp * q[j] * L.d[3 * p + 2];
        }
      }
    nbr_free(&L);
    return;
  }
  for (int i = 0; i < nlocal; i++) {
    double qtmp = q[i], xtmp = x[3 * i], ytmp = x[3 * i + 1], ztmp = x[3 * i + 2];
    for (int j = i + 1; j < nlocal; j++) {
      double xjimage[3];
      orc_closest_image(s, &x[3 * i], &x[3 * j], xjimage);
      double delx = xtmp - xjimage[0], dely = ytmp - xjimage[1], delz = ztmp - xjimage[2];
      double rsq = delx * delx + dely * dely + delz * delz;
      if (rsq <= cut_coulsq) {
        if ((mol[i] != mol[j]) || mol[i] == 0) {
          double r = sqrt(rsq);
          double dvdrr = 1.0 / rsq + f_shift;
          double ef_temp = dvdrr * 1.0 / r;
          ef_static[3 * i] += ef_temp * q[j] * delx;
          ef_static[3 * i + 1] += ef_temp * q[j] * dely;
          ef_static[3 * i + 2] += ef_temp * q[j] * delz;
          ef_static[3 * j] -= ef_temp * qtmp * delx;
          ef_static[3 * j + 1] -= ef_temp * qtmp * dely;
          ef_static[3 * j + 2] -= ef_temp * qtmp * delz;
        }
      }
    }
  }
}

/* ------------------------------------------------------------------ a6 --
 * PS.cpp:1243-1316: dense 3N x 3N dipole field tensor, no cutoff.           */
static void tensor_block(const orc_system *s, const double *d, double r2, double *T /*[9]*/) {
  /* PS.cpp:1284-1306 */
  double r = sqrt(r2), r3, r5, damping_term1 = 1.0, damping_term2 = 1.0;
  const double pd = s->polar_damp;
  if (r == 0.0) r3 = r5 = DBL_MAX;
  else { r3 = 1.0 / (r * r * r); r5 = 1.0 / (r * r * r * r * r); }
  if (s->damping_type == ORC_DAMP_EXPONENTIAL) {
    damping_term1 = 1.0 - exp(-pd * r) * (0.5 * pd * pd * r2 + pd * r + 1.0);
    damping_term2 = 1.0 - exp(-pd * r) * (pd * pd * pd * r2 * r / 6.0 + 0.5 * pd * pd * r2 + pd * r + 1.0);
  }
  for (int p = 0; p < 3; p++)
    for (int qq = 0; qq < 3; qq++) {
      T[3 * p + qq] = -3.0 * d[p] * d[qq] * damping_term2 * r5;
      if (p == qq) T[3 * p + qq] += damping_term1 * r3;
    }
}

void orc_build_dipole_field_matrix(const orc_system *s, double *M) {
  const int N = s->nlocal;
  const size_t ld = 3 * (size_t)N;
  const double *x = s->x;
  memset(M, 0, sizeof(double) * ld * ld);
  for (int i = 0; i < N; i++)
    for (int p = 0; p < 3; p++)
      M[(3 * (size_t)i + p) * ld + 3 * i + p] = (s->alpha[i] != 0.0) ? 1.0 / s->alpha[i] : DBL_MAX;
  for (int i = 0; i < N - 1; i++)
    for (int j = i + 1; j < N; j++) {
      double xjimage[3], d[3], T[9];
      orc_closest_image(s, &x[3 * i], &x[3 * j], xjimage);
      d[0] = x[3 * i] - xjimage[0]; d[1] = x[3 * i + 1] - xjimage[1]; d[2] = x[3 * i + 2] - xjimage[2];
      double r2 = pow(d[0], 2) + pow(d[1], 2) + pow(d[2], 2);
      tensor_block(s, d, r2, T);
      for (int p = 0; p < 3; p++)
        for (int qq = 0; qq < 3; qq++) {
          M[(3 * (size_t)i + p) * ld + 3 * j + qq] = T[3 * p + qq];
          M[(3 * (size_t)j + p) * ld + 3 * i + qq] = T[3 * p + qq]; /* PS.cpp:1309-1311 */
        }
    }
}

/* stable descending order by rank_metric == the reference's bubble sort (PS.cpp:1130-1143):
 * adjacent swaps only when strictly smaller, so equal keys keep index order. */
static const double *g_sort_key;
static int cmp_rank(const void *a, const void *b) {
  int ia = *(const int *)a, ib = *(const int *)b;
  double ka = g_sort_key[ia], kb = g_sort_key[ib];
  if (ka > kb) return -1;
  if (ka < kb) return 1;
  return (ia > ib) - (ia < ib);
}

/* ------------------------------------------------------------------ a7 --
 * PS.cpp:1113-1238.  `matrix` dense (reference semantics) or NULL with dd_cutoff>0, in which
 * case the sparse tensor list built by orc_compute is passed through `res`-side statics.      */
/* tilted boxes (list-mode extension in a triclinic cell; test systems are small): every pair through
 * Domain::closest_image's triclinic branch (orc_closest_image), no grid */
static void nbr_build_allpairs(const orc_system *s, double cut, nbr_list *L) {
  const int n = s->nlocal;
  const double *x = s->x;
  const double cutsq = cut * cut;
  L->first = (long long *)malloc(sizeof(long long) * ((size_t)n + 1));
  for (int pass = 0; pass < 2; pass++) {
    long long tot = 0;
    for (int i = 0; i < n; i++) {
      if (pass == 1) tot = L->first[i];
      for (int j = 0; j < n; j++) {
        if (j == i) continue;
        double xim[3];
        orc_closest_image(s, &x[3 * i], &x[3 * j], xim);
        double ddx = x[3 * i] - xim[0], ddy = x[3 * i + 1] - xim[1], ddz = x[3 * i + 2] - xim[2];
        double rsq = ddx * ddx + ddy * ddy + ddz * ddz;
        if (rsq <= cutsq) {
          if (pass == 1) {
            L->j[tot] = j; L->rsq[tot] = rsq;
            L->d[3 * tot] = ddx; L->d[3 * tot + 1] = ddy; L->d[3 * tot + 2] = ddz;
          }
          tot++;
        }
      }
      if (pass == 0) L->first[i + 1] = tot;
    }
    if (pass == 0) {
      L->first[0] = 0;
      L->npairs = L->first[n];
      L->j = (int *)malloc(sizeof(int) * (size_t)(L->npairs + 1));
      L->d = (double *)malloc(sizeof(double) * 3 * (size_t)(L->npairs + 1));
      L->rsq = (double *)malloc(sizeof(double) * (size_t)(L->npairs + 1));
    }
  }
}

typedef struct { const nbr_list *L; const double *T6; } sparse_T; /* T6: xx,xy,xz,yy,yz,zz per pair */
static const sparse_T *g_sparse = NULL;

int orc_dipole_solver(const orc_system *s, const double *M, const double *ef_static,
                      const double *rank_metric, double *mu, orc_result *res, double *utrace) {
  const int nlocal = s->nlocal;
  const size_t ld = 3 * (size_t)nlocal;
  const double *alpha = s->alpha;
  double *mu_new = (double *)calloc(ld + 3, sizeof(double));
  double *mu_old = (double *)calloc(ld + 3, sizeof(double));
  double *ef_ind = (double *)calloc(ld + 3, sizeof(double));
  int *ranked = (int *)malloc(sizeof(int) * ((size_t)nlocal + 1));
  int keep_iterating = 1, iterations = 0;
  res->status = 0; res->sweeps = 0; res->rms_dmu = 0.0;
  for (int i = 0; i < nlocal; i++) ranked[i] = i;
  if (s->polar_gs_ranked) {
    g_sort_key = rank_metric;
    qsort(ranked, (size_t)nlocal, sizeof(int), cmp_rank);
  }
  while (keep_iterating) {
    for (size_t k = 0; k < ld; k++) { mu_old[k] = mu[k]; ef_ind[k] = 0; }
    for (int i = 0; i < nlocal; i++) {
      int index = ranked[i];
      size_t ii = 3 * (size_t)index;
      if (M) {
        for (int j = 0; j < nlocal; j++) {
          size_t jj = 3 * (size_t)j;
          if (index != j)
            for (int p = 0; p < 3; p++)
              for (int qq = 0; qq < 3; qq++) ef_ind[ii + p] -= M[(ii + p) * ld + jj + qq] * mu[jj + qq];
        }
      } else {
        const nbr_list *L = g_sparse->L;
        const double *T6 = g_sparse->T6;
        for (long long pp = L->first[index]; pp < L->first[index + 1]; pp++) {
          size_t jj = 3 * (size_t)L->j[pp];
          const double *T = T6 + 6 * pp;
          ef_ind[ii] -= T[0] * mu[jj]; ef_ind[ii] -= T[1] * mu[jj + 1]; ef_ind[ii] -= T[2] * mu[jj + 2];
          ef_ind[ii + 1] -= T[1] * mu[jj]; ef_ind[ii + 1] -= T[3] * mu[jj + 1]; ef_ind[ii + 1] -= T[4] * mu[jj + 2];
          ef_ind[ii + 2] -= T[2] * mu[jj]; ef_ind[ii + 2] -= T[4] * mu[jj + 1]; ef_ind[ii + 2] -= T[5] * mu[jj + 2];
        }
      }
      for (int p = 0; p < 3; p++) {
        mu_new[ii + p] = alpha[index] * (ef_static[ii + p] + ef_ind[ii + p]);
        if (s->polar_gs || s->polar_gs_ranked) mu[ii + p] = mu_new[ii + p];
      }
    }
    res->sweeps++;
    if (utrace) { /* PS.cpp:1182-1191 (value before the K conversion factor) */
      double u = 0.0;
      for (size_t k = 0; k < ld; k++) u += ef_static[k] * mu[k];
      utrace[iterations] = -0.5 * u;
    }
    if (s->fixed_iteration == 0) {
      keep_iterating = 0;
      double change = 0;
      for (size_t k = 0; k < ld; k++) change += (mu_new[k] - mu_old[k]) * (mu_new[k] - mu_old[k]);
      change /= (double)(nlocal) * 3.0;
      res->rms_dmu = sqrt(change);
      if (change > s->polar_precision * s->polar_precision) keep_iterating = 1;
    } else {
      if (iterations >= s->iterations_max) goto done; /* PS.cpp:1214: returns BEFORE the copy */
    }
    for (size_t k = 0; k < ld; k++) mu[k] = mu_new[k];
    iterations++;
    if (iterations > s->iterations_max) { /* PS.cpp:1227-1235 */
      for (size_t k = 0; k < ld; k++) mu[k] = alpha[k / 3] * ef_static[k];
      res->status = 1;
      goto done;
    }
  }
done:
  free(mu_new); free(mu_old); free(ef_ind); free(ranked);
  res->iterations = iterations;
  return iterations;
}

/* ------------------------------------------------------------------ a8 --
 * PS.cpp:406-641 pair body, shared by the all-pairs and the cell-list drivers.
 * Returns the force on i (to be subtracted from j).                         */
static void polar_pair(const orc_system *s, int eflag, int i, int j, double delx, double dely,
                       double delz, const double *mu, double cut_coulsq, double f_shift, double e2s,
                       double ddcutsq, double *fout, double *u_ef, double *u_dd, double *fdd /* dipole-dipole part of fout */) {
  const double *q = s->q, *alpha = s->alpha;
  const int *mol = s->molecule;
  const double pd = s->polar_damp;
  double qtmp = q[i];
  double xsq = delx * delx, ysq = dely * dely, zsq = delz * delz;
  double rsq = xsq + ysq + zsq;
  double r2inv = 1.0 / rsq;
  double rinv = sqrt(r2inv);
  double r = 1.0 / rinv;
  double r3inv = r2inv * rinv;
  double fx = 0.0, fy = 0.0, fz = 0.0;
  fdd[0] = fdd[1] = fdd[2] = 0.0;
  const double *mi = mu + 3 * (size_t)i, *mj = mu + 3 * (size_t)j;

  if (rsq < cut_coulsq) {
    if ((mol[i] != mol[j]) || mol[i] == 0) {
      double dvdrr = 1.0 / rsq + f_shift;
      double ef_temp = dvdrr * 1.0 / r * e2s;
      if (alpha[i] != 0.0 && q[j] != 0.0) { /* dipole on i, charge on j */
        double cf = q[j] * e2s * r3inv;
        fx += cf * (mi[0] * ((-2.0 * xsq + ysq + zsq) * r2inv + f_shift * (ysq + zsq)) +
                    mi[1] * (-3.0 * delx * dely * r2inv - f_shift * delx * dely) +
                    mi[2] * (-3.0 * delx * delz * r2inv - f_shift * delx * delz));
        fy += cf * (mi[0] * (-3.0 * delx * dely * r2inv - f_shift * delx * dely) +
                    mi[1] * ((-2.0 * ysq + xsq + zsq) * r2inv + f_shift * (xsq + zsq)) +
                    mi[2] * (-3.0 * dely * delz * r2inv - f_shift * dely * delz));
        fz += cf * (mi[0] * (-3.0 * delx * delz * r2inv - f_shift * delx * delz) +
                    mi[1] * (-3.0 * dely * delz * r2inv - f_shift * dely * delz) +
                    mi[2] * ((-2.0 * zsq + xsq + ysq) * r2inv + f_shift * (xsq + ysq)));
        if (eflag) {
          double e0 = ef_temp * q[j] * delx, e1 = ef_temp * q[j] * dely, e2 = ef_temp * q[j] * delz;
          *u_ef -= mi[0] * e0 + mi[1] * e1 + mi[2] * e2;
        }
      }
      if (alpha[j] != 0.0 && qtmp != 0.0) { /* dipole on j, charge on i */
        double cf = qtmp * e2s * r3inv;
        fx -= cf * (mj[0] * ((-2.0 * xsq + ysq + zsq) * r2inv + f_shift * (ysq + zsq)) +
                    mj[1] * (-3.0 * delx * dely * r2inv - f_shift * delx * dely) +
                    mj[2] * (-3.0 * delx * delz * r2inv - f_shift * delx * delz));
        fy -= cf * (mj[0] * (-3.0 * delx * dely * r2inv - f_shift * delx * dely) +
                    mj[1] * ((-2.0 * ysq + xsq + zsq) * r2inv + f_shift * (xsq + zsq)) +
                    mj[2] * (-3.0 * dely * delz * r2inv - f_shift * dely * delz));
        fz -= cf * (mj[0] * (-3.0 * delx * delz * r2inv - f_shift * delx * delz) +
                    mj[1] * (-3.0 * dely * delz * r2inv - f_shift * dely * delz) +
                    mj[2] * ((-2.0 * zsq + xsq + ysq) * r2inv + f_shift * (xsq + ysq)));
        if (eflag) {
          double e0 = ef_temp * qtmp * delx, e1 = ef_temp * qtmp * dely, e2 = ef_temp * qtmp * delz;
          *u_ef += mj[0] * e0 + mj[1] * e1 + mj[2] * e2;
        }
      }
    }
  }
  /* dipole-dipole: no cutoff, no molecule exclusion in the reference (PS.cpp:512);
   * ddcutsq < 0 means "none"; the extension truncates at rsq < ddcutsq. */
  if (alpha[i] != 0.0 && alpha[j] != 0.0 && (ddcutsq < 0.0 || rsq < ddcutsq)) {
    double r5inv = r3inv * r2inv, r7inv = r5inv * r2inv;
    double pdotp = mi[0] * mj[0] + mi[1] * mj[1] + mi[2] * mj[2];
    double pidotr = mi[0] * delx + mi[1] * dely + mi[2] * delz;
    double pjdotr = mj[0] * delx + mj[1] * dely + mj[2] * delz;
    if (s->damping_type == ORC_DAMP_EXPONENTIAL) {
      double term_1 = exp(-pd * r);
      double term_2 = 1.0 + pd * r + 0.5 * pd * pd * r * r;
      double term_3 = 1.0 + pd * r + 0.5 * pd * pd * r * r + 1.0 / 6.0 * pd * pd * pd * r * r * r;
      double pre1 = 3.0 * r5inv * pdotp * (1.0 - term_1 * term_2) -
                    15.0 * r7inv * pidotr * pjdotr * (1.0 - term_1 * term_3);
      double pre2 = 3.0 * r5inv * pjdotr * (1.0 - term_1 * term_3);
      double pre3 = 3.0 * r5inv * pidotr * (1.0 - term_1 * term_3);
      double pre4 = -pdotp * r3inv * (-term_1 * (pd * rinv + pd * pd) + term_1 * pd * term_2 * rinv);
      double pre5 = 3.0 * pidotr * pjdotr * r5inv *
                    (-term_1 * (pd * rinv + pd * pd + 0.5 * r * pd * pd * pd) + term_1 * pd * term_3 * rinv);
      fdd[0] = pre1 * delx + pre2 * mi[0] + pre3 * mj[0] + pre4 * delx + pre5 * delx;   /* PS.cpp:544-550 */
      fdd[1] = pre1 * dely + pre2 * mi[1] + pre3 * mj[1] + pre4 * dely + pre5 * dely;
      fdd[2] = pre1 * delz + pre2 * mi[2] + pre3 * mj[2] + pre4 * delz + pre5 * delz;
      fx += fdd[0]; fy += fdd[1]; fz += fdd[2];
      if (eflag)
        *u_dd += r3inv * pdotp * (1.0 - term_1 * term_2) - 3.0 * r5inv * pidotr * pjdotr * (1.0 - term_1 * term_3);
    } else {
      double pre1 = 3.0 * r5inv * pdotp - 15.0 * r7inv * pidotr * pjdotr;
      double pre2 = 3.0 * r5inv * pjdotr;
      double pre3 = 3.0 * r5inv * pidotr;
      fdd[0] = pre1 * delx + pre2 * mi[0] + pre3 * mj[0];   /* PS.cpp:585-591 */
      fdd[1] = pre1 * dely + pre2 * mi[1] + pre3 * mj[1];
      fdd[2] = pre1 * delz + pre2 * mi[2] + pre3 * mj[2];
      fx += fdd[0]; fy += fdd[1]; fz += fdd[2];
      if (eflag) *u_dd += r3inv * pdotp - 3.0 * r5inv * pidotr * pjdotr;
    }
  }
  fout[0] = fx; fout[1] = fy; fout[2] = fz;
}

/* per-atom part of ev_tally_xyz, pair.cpp:1065-1082 (newton or both local: every pair here) */
static void vatom_xyz(double *vatom, int i, int j, const double *d, const double *fo) {
  double v[6] = {d[0] * fo[0], d[1] * fo[1], d[2] * fo[2], d[0] * fo[1], d[0] * fo[2], d[1] * fo[2]};
  for (int k = 0; k < 6; k++) { vatom[6 * i + k] += 0.5 * v[k]; vatom[6 * j + k] += 0.5 * v[k]; }
}

void orc_polar_forces(const orc_system *s, int eflag, int vflag_pairwise, const double *mu, double *f,
                      orc_result *res, double *vatom) {
  const int nlocal = s->nlocal;
  const double *x = s->x, *alpha = s->alpha;
  const double cut_coulsq = s->cut_coul * s->cut_coul;
  const double f_shift = -1.0 / (s->cut_coul * s->cut_coul);
  const double e2s = sqrt(s->qqrd2e);
  double u_self = 0.0, u_ef = 0.0, u_dd = 0.0;
  if (s->dd_cutoff > 0.0) {
    nbr_list L; memset(&L, 0, sizeof(L));
    double cut = s->dd_cutoff > s->cut_coul ? s->dd_cutoff : s->cut_coul;
    nbr_build(s, cut, &L);
    double ddsq = s->dd_cutoff * s->dd_cutoff;
    for (int i = 0; i < nlocal; i++) {
      if (eflag && alpha[i] != 0.0)
        u_self += 0.5 * (mu[3 * i] * mu[3 * i] + mu[3 * i + 1] * mu[3 * i + 1] + mu[3 * i + 2] * mu[3 * i + 2]) / alpha[i];
      for (long long p = L.first[i]; p < L.first[i + 1]; p++) {
        int j = L.j[p];
        if (j < i) continue; /* i<j, same orientation as the reference */
        double fo[3], fdd[3];
        polar_pair(s, eflag, i, j, L.d[3 * p], L.d[3 * p + 1], L.d[3 * p + 2], mu, cut_coulsq, f_shift, e2s,
                   ddsq, fo, &u_ef, &u_dd, fdd);
        for (int k = 0; k < 3; k++) { f[3 * i + k] += fo[k]; f[3 * j + k] -= fo[k]; }
        if (i == 0) for (int k = 0; k < 3; k++) { res->force_atom0[k] += fo[k]; res->dipole_force_atom0[k] += fdd[k]; }   /* PS.cpp:546-550, 616-620 */
        if (j == 0) for (int k = 0; k < 3; k++) { res->force_atom0[k] -= fo[k]; res->dipole_force_atom0[k] -= fdd[k]; }   /* PS.cpp:552-556, 622-626 */
        if (vflag_pairwise) {
          const double *d = &L.d[3 * p];
          res->virial[0] += d[0] * fo[0]; res->virial[1] += d[1] * fo[1]; res->virial[2] += d[2] * fo[2];
          res->virial[3] += d[0] * fo[1]; res->virial[4] += d[0] * fo[2]; res->virial[5] += d[1] * fo[2];
        }
        if (vatom) vatom_xyz(vatom, i, j, &L.d[3 * p], fo);
      }
    }
    nbr_free(&L);
  } else {
    for (int i = 0; i < nlocal; i++) {
      double xtmp = x[3 * i], ytmp = x[3 * i + 1], ztmp = x[3 * i + 2];
      if (eflag && alpha[i] != 0.0)
        u_self += 0.5 * (mu[3 * i] * mu[3 * i] + mu[3 * i + 1] * mu[3 * i + 1] + mu[3 * i + 2] * mu[3 * i + 2]) / alpha[i];
      for (int j = i + 1; j < nlocal; j++) {
        double xjimage[3], fo[3], fdd[3];
        orc_closest_image(s, &x[3 * i], &x[3 * j], xjimage);
        double delx = xtmp - xjimage[0], dely = ytmp - xjimage[1], delz = ztmp - xjimage[2];
        polar_pair(s, eflag, i, j, delx, dely, delz, mu, cut_coulsq, f_shift, e2s, -1.0, fo, &u_ef, &u_dd, fdd);
        for (int k = 0; k < 3; k++) { f[3 * i + k] += fo[k]; f[3 * j + k] -= fo[k]; }
        if (i == 0) for (int k = 0; k < 3; k++) { res->force_atom0[k] += fo[k]; res->dipole_force_atom0[k] += fdd[k]; }
        if (j == 0) for (int k = 0; k < 3; k++) { res->force_atom0[k] -= fo[k]; res->dipole_force_atom0[k] -= fdd[k]; }
        if (vflag_pairwise) { /* ev_tally_xyz pair.cpp:1001-1075, newton or both local */
          res->virial[0] += delx * fo[0]; res->virial[1] += dely * fo[1]; res->virial[2] += delz * fo[2];
          res->virial[3] += delx * fo[1]; res->virial[4] += delx * fo[2]; res->virial[5] += dely * fo[2];
        }
        if (vatom) { double d[3] = {delx, dely, delz}; vatom_xyz(vatom, i, j, d, fo); }
      }
    }
  }
  res->u_self = u_self; res->u_ef = u_ef; res->u_dd = u_dd;
  res->eng_pol = u_self + u_ef + u_dd; /* PS.cpp:632,641 (all zero when !eflag) */
}

/* ----------------------------------------------------------------- a10 -- */
void orc_virial_fdotr(const orc_system *s, const double *f, double *virial) {
  const int nall = s->nlocal + s->nghost;
  const double *x = s->x;
  for (int i = 0; i < nall; i++) {
    virial[0] += f[3 * i] * x[3 * i];
    virial[1] += f[3 * i + 1] * x[3 * i + 1];
    virial[2] += f[3 * i + 2] * x[3 * i + 2];
    virial[3] += f[3 * i + 1] * x[3 * i];
    virial[4] += f[3 * i + 2] * x[3 * i];
    virial[5] += f[3 * i + 2] * x[3 * i + 1];
  }
}

/* --------------------------------------------------------- compute() --
 * PS.cpp:125-645 in order.                                               */
int orc_compute(const orc_system *s, int eflag, int vflag, double *f, double *mu, double *ef_static,
                orc_result *res, double *utrace) {
  return orc_compute_peratom(s, eflag, vflag, f, mu, ef_static, res, utrace, NULL, NULL);
}

/* eflag/2 -> eatom[nall], vflag/4 -> vatom[nall][6] (pair.cpp:760-764); the caller zeroes them
 * like ev_setup does (pair.cpp:789-806).                                                    */
int orc_compute_peratom(const orc_system *s, int eflag, int vflag, double *f, double *mu, double *ef_static,
                        orc_result *res, double *utrace, double *eatom, double *vatom) {
  const int nlocal = s->nlocal;
  double t0;
  memset(res, 0, sizeof(*res));
  double *rank_metric = (double *)calloc((size_t)nlocal + 1, sizeof(double));
  const int vpair = (vflag % 4) == 1;

  t0 = now_s();
  if (s->polar_gs_ranked) orc_rank_metric(s, rank_metric, &res->rmin); /* a2 */
  res->t_rank = now_s() - t0;

  t0 = now_s();
  if (!(eflag / 2)) eatom = NULL;
  if (!(vflag / 4)) vatom = NULL;
  orc_ljcoul(s, eflag, vpair, f, &res->eng_vdwl, &res->eng_coul, res->virial, eatom, vatom); /* a3 */
  res->t_ljcoul = now_s() - t0;

  t0 = now_s();
  orc_static_field(s, ef_static); /* a4 */
  const double e2s = sqrt(s->qqrd2e);
  for (int i = 0; i < nlocal; i++) /* a5: PS.cpp:363-386 */
    for (int k = 0; k < 3; k++) {
      ef_static[3 * i + k] = ef_static[3 * i + k] * e2s;
      if (!s->use_previous) {
        mu[3 * i + k] = s->alpha[i] * ef_static[3 * i + k];
        mu[3 * i + k] *= s->polar_gamma;
      }
    }
  res->t_static = now_s() - t0;

  int iterations = 0;
  if (!s->zodid) { /* PS.cpp:389 */
    if (s->dd_cutoff > 0.0) {
      /* extension: sparse tensor list (rsq < dd_cutoff^2), canonical i<j orientation */
      t0 = now_s();
      nbr_list L; memset(&L, 0, sizeof(L));
      nbr_build(s, s->dd_cutoff, &L);
      double *T6 = (double *)malloc(sizeof(double) * 6 * (size_t)(L.npairs + 1));
      const double ddsq = s->dd_cutoff * s->dd_cutoff;
      for (int i = 0; i < nlocal; i++)
        for (long long p = L.first[i]; p < L.first[i + 1]; p++) {
          int j = L.j[p];
          double d[3] = {L.d[3 * p], L.d[3 * p + 1], L.d[3 * p + 2]}, T[9];
          if (j < i) { d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2]; }
          double r2 = pow(d[0], 2) + pow(d[1], 2) + pow(d[2], 2);
          if (r2 < ddsq) tensor_block(s, d, r2, T); else memset(T, 0, sizeof(T));
          T6[6 * p] = T[0]; T6[6 * p + 1] = T[1]; T6[6 * p + 2] = T[2];
          T6[6 * p + 3] = T[4]; T6[6 * p + 4] = T[5]; T6[6 * p + 5] = T[8];
        }
      res->t_matrix = now_s() - t0;
      sparse_T sp = {&L, T6};
      g_sparse = &sp;
      t0 = now_s();
      iterations = orc_dipole_solver(s, NULL, ef_static, rank_metric, mu, res, utrace);
      res->t_solve = now_s() - t0;
      g_sparse = NULL;
      free(T6);
      nbr_free(&L);
    } else {
      t0 = now_s();
      size_t ld = 3 * (size_t)nlocal;
      double *M = (double *)malloc(sizeof(double) * ld * ld + 8);
      if (!M) { free(rank_metric); return -1; }
      orc_build_dipole_field_matrix(s, M); /* a6 */
      res->t_matrix = now_s() - t0;
      t0 = now_s();
      iterations = orc_dipole_solver(s, M, ef_static, rank_metric, mu, res, utrace); /* a7 */
      res->t_solve = now_s() - t0;
      free(M);
    }
  }
  res->iterations = iterations;

  t0 = now_s();
  orc_polar_forces(s, eflag, vpair, mu, f, res, vatom); /* a8 */
  res->t_force = now_s() - t0;

  if ((vflag % 4) == 2) orc_virial_fdotr(s, f, res->virial); /* a10; f must hold pair forces only */
  free(rank_metric);
  return res->status;
}
