/*
 * bdf_core.h -- the CVODES controller, ONE source for every register-resident mapping.
 *
 * Everything of the restated CVODES 5.x step that does not depend on where an instance's vectors live: cvHin,
 * Nordsieck rescale / order change / predict / restore, the Newton control loop (SUNNonlinSol_Newton semantics,
 * simultaneous and staggered sensitivity correctors), error tests and their failure handling, cvCompleteStep,
 * cvPrepareNextStep (order selection as one straight-line block), CVodeGetDky, the first-call block, the per-attempt
 * driver (cv_attempt).  Included by
 *   bdf_kernels.hip   one lane per instance (n <= 5): a vector = NS doubles of one lane,
 *   bdf_wave.hip      G lanes per instance / a workgroup per instance: a vector = RS register slots per lane,
 * after each has defined its MAPPING -- the only things that differ between them:
 *   SA_STATE<BWD>     the state struct (fields named as in CVodeMem: zn, znQ, acor, ewt, tq, l, tau, ...); a vector
 *                     field is anything indexable as v[r] (register arrays, or views of an HBM workspace),
 *   RS, RQ            doubles of a state / quadrature vector held by one lane;  IDX(m, r) = component of slot r
 *   VFOR(r) .. VEND / QFOR(r) .. QEND    loop over the slots of a state / quadrature vector: compile-time
 *                     (SFOR over RS / RQ, the default in sa_common.h) or a run-time loop (bdf_mem.hip),
 *   wrms_n / wrms_q / quad_update_norm / ewt_set / ewtQ_set / wave_max    norms and reductions (in-lane trees or
 *                     cross-lane butterflies: the association is the oracle's balanced tree in both)
 *   cv_f / cv_fQ / cv_jac, cv_fS     generated callbacks (registers, or staged through LDS)
 *   cv_lsetup / dense_getrs           Newton matrix set-up + LU, triangular solves
 *   interp_y          forward state at t from the stored trajectory (backward problem)
 *   SV / SLOOP_BEGIN / SLOOP_END      sensitivity vectors (registers, or streamed from the workspace)
 *   COLD_STORE / COLD_LOAD, PH_T0 / PH_ADD           optional hooks (LDS parking of cold state, phase timers)
 *   TMPV / TMPQ / TMPV2                              optional: where the controller's vector temporaries live
 *   SA_POLY_CM(BWD)   whether the pow polynomials read their coefficients from constant memory (sa_common.h)
 * A controller change is an edit of THIS file; bit-equality with the oracle (tests -m gpu) covers every mapping.
 * Since round 5 also included by
 *   bdf_mem.hip       one lane per instance, every vector a strided view of an HBM workspace, run-time loops
 *                     over the components (n > 128, and forward sensitivities beyond the lane groups).
 *
 * Reference call sites: /root/reference/sunode/solver.py:467-527, 682-784; CVODES digest: SURVEY.md Appendix A / B.
 */
#ifndef SA_BDF_CORE_H
#define SA_BDF_CORE_H

/* state- / quadrature-sized temporaries of the controller: per-lane register arrays, unless the mapping keeps its
   vectors elsewhere (bdf_mem.hip: numbered slots of the HBM workspace -- an n-sized array per lane would be
   O(n) scratch per lane there).  Slots live at the same time never share a number. */
#ifndef TMPV
#define TMPV(m, name, slot) double name[RS]
#define TMPQ(m, name, slot) double name[RQ]
#define TMPV2(m, name, rows, slot) double name[rows][RS]
#endif
#define SA_TMPV_SLOTS 20
#define SA_TMPQ_SLOTS 5

#ifdef SA_SENS
/* ---- forward sensitivities: the parts of the corrector that do not depend on the mapping ---- */
/* cvSensEwtSetEE: w[is] = pbar / (rtol |pbar s| + atol) */
template <int v_in, int v_out, bool BWD>
DEV int sens_ewt_set(SA_STATE<BWD> &m)
{
    double bad = 0.0;
    SLOOP_BEGIN(is)
        const double pb = m.pbar[is];
        VFOR(r) {
            const double v = FMA(m.rtol, fabs(pb * SV(m, v_in, is, r)), m.atol[r]);
            bad = (IDX(m, r) < NS && v <= 0.0) ? 1.0 : bad;
            SV(m, v_out, is, r) = pb * (1.0 / v);
        } VEND
    SLOOP_END
    return wave_max(m.lane, bad) > 0.0 ? -1 : 0;
}

/* cvSensUpdateNorm: max(old, max_is wrms(x[is], w[is])) */
template <int v_x, int v_w, bool BWD>
DEV double sens_update_norm(const SA_STATE<BWD> &m, double old_nrm)
{
    double nrm = old_nrm;
    SLOOP_BEGIN(is)
        TMPV(m, x, 0); TMPV(m, w, 1);
        VFOR(r) { x[r] = SV(m, v_x, is, r); w[r] = SV(m, v_w, is, r); } VEND
        const double snrm = wrms_n(m, x, w);
        nrm = snrm > nrm ? snrm : nrm;
    SLOOP_END
    return nrm;
}

/* cvNlsResidualSensSim / ...Stg: residuals of the sensitivity systems -> DELTA (m.y holds the state) */
template <bool BWD>
DEV int cv_nls_residual_sens(SA_STATE<BWD> &m)
{
    SLOOP_BEGIN(is) VFOR(r) SV(m, SV_Y, is, r) = SV(m, SV_ZN0, is, r) + SV(m, SV_ACOR, is, r); VEND SLOOP_END
    int retval = cv_fS<SV_Y, SV_FTEMP>(m, m.tn, m.y);
    if (retval < 0) return CV_SRHSFUNC_FAIL;
    if (retval > 0) return SRHSFUNC_RECVR;
    SLOOP_BEGIN(is)
        VFOR(r) {
            const double rr = FMA(m.rl1, SV(m, SV_ZN0 + 1, is, r), SV(m, SV_ACOR, is, r));
            SV(m, SV_DELTA, is, r) = FMA(-m.gamma, SV(m, SV_FTEMP, is, r), rr);
        } VEND
    SLOOP_END
    return CV_SUCCESS;
}

/* one Newton update of every sensitivity system with the current factorisation */
template <bool BWD>
DEV void cv_sens_newton_update(SA_STATE<BWD> &m)
{
    SLOOP_BEGIN(is)
        TMPV(m, d, 2);
        VFOR(r) d[r] = -1.0 * SV(m, SV_DELTA, is, r); VEND
        dense_getrs(m, d);
        if (m.gamrat != 1.0) {
            double sc = 2.0 / (1.0 + m.gamrat);
            VFOR(r) d[r] *= sc; VEND
        }
        VFOR(r) { SV(m, SV_DELTA, is, r) = d[r]; SV(m, SV_ACOR, is, r) = SV(m, SV_ACOR, is, r) + d[r]; } VEND
    SLOOP_END
}
#endif

/* ---- CVodeInit / CVodeReInit ---- */
template <bool BWD, class VY, class VQ>
DEV void cv_reinit(SA_STATE<BWD> &m, double t0, const VY &y0, const VQ &q0)
{
    m.tn = t0;
    m.q = 1; m.L = 2; m.qwait = 2; m.etamax = ETAMX1;
    m.qu = 0; m.hu = 0.0;
    SFOR(j, 0, (QMAX) + 1) {
        VFOR(r) m.zn[j][r] = 0.0; VEND
        QFOR(r) m.znQ[j][r] = 0.0; QEND
    } SEND
    VFOR(r) m.zn[0][r] = y0[r]; VEND
    if (BWD) { QFOR(r) m.znQ[0][r] = q0[r]; QEND }
    m.nst = m.nfe = m.ncfn = m.netf = m.nni = m.nsetups = 0;
    m.nje = 0; m.nstlp = 0; m.nstlj = 0; m.nfQe = m.netfQ = 0;
    m.h = 0.0; m.hprime = 0.0; m.hscale = 0.0; m.eta = 1.0;
    m.qprime = 1;
    m.gamma = m.gammap = 0.0; m.gamrat = 1.0; m.crate = 1.0; m.delp = 0.0;
    m.acnrm = 0.0; m.saved_tq5 = 0.0;
    m.jcur = 0; m.nls_jcur = 0;
    SFOR(i, 0, 7) { m.tau[i] = 0.0; m.l[i] = 0.0; } SEND
    SFOR(i, 0, 6) m.tq[i] = 0.0; SEND
    VFOR(r) { m.acor[r] = m.tempv[r] = m.ftemp[r] = m.y[r] = m.zsave[r] = 0.0; } VEND
    QFOR(r) { m.acorQ[r] = m.tempvQ[r] = m.zsaveQ[r] = 0.0; } QEND
#ifdef SA_SENS
    m.crateS = 1.0; m.delpS = 0.0; m.acnrmS = 0.0;
    m.nfSe = m.nniS = m.ncfnS = m.netfS = m.nsetupsS = 0;
#endif
}

/* ---- cvHin ---- */
template <bool BWD>
DEV double cv_upper_bound_h0(SA_STATE<BWD> &m, double tdist)
{
    TMPV(m, w, 3);
    ewt_set(m, m.zn[0], w);
    double loc = 0.0;
    VFOR(r) {
        const double t1 = FMA(HUB_FACTOR, fabs(m.zn[0][r]), 1.0 / w[r]);
        const double v = (IDX(m, r) < NS) ? fabs(m.zn[1][r]) / t1 : 0.0;
        loc = v > loc ? v : loc;
    } VEND
    double hub_inv = wave_max(m.lane, loc);
#ifdef SA_SENS
    if (SENS_ON(m)) {
        sens_ewt_set<SV_ZN0, SV_TEMPV>(m);
        double locS = 0.0;
        SLOOP_BEGIN(is)
            VFOR(r) {
                const double t2 = fabs(SV(m, SV_ZN0, is, r));
                double t1 = 1.0 / SV(m, SV_TEMPV, is, r);
                t1 = FMA(HUB_FACTOR, t2, t1);
                const double v = (IDX(m, r) < NS) ? fabs(SV(m, SV_ZN0 + 1, is, r)) / t1 : 0.0;
                locS = v > locS ? v : locS;
            } VEND
        SLOOP_END
        const double hubS = wave_max(m.lane, locS);
        if (hubS > hub_inv) hub_inv = hubS;
    }
#endif
    if (BWD) {
        TMPQ(m, wq, 4);
        ewtQ_set(m, m.znQ[0], wq);
        double locq = 0.0;
        QFOR(r) {
            const double t1q = FMA(HUB_FACTOR, fabs(m.znQ[0][r]), 1.0 / wq[r]);
            const double v = (IDX(m, r) < NQ) ? fabs(m.znQ[1][r]) / t1q : 0.0;
            locq = v > locq ? v : locq;
        } QEND
        const double hubQ_inv = wave_max(m.lane, locq);
        if (hubQ_inv > hub_inv) hub_inv = hubQ_inv;
    }
    double hub = HUB_FACTOR * tdist;
    if (hub * hub_inv > 1.0) hub = 1.0 / hub_inv;
    return hub;
}

template <bool BWD>
DEV int cv_ydd_norm(SA_STATE<BWD> &m, double hg, double *yddnrm)
{
    VFOR(r) m.y[r] = FMA(hg, m.zn[1][r], m.zn[0][r]); VEND
#ifdef SA_SENS
    if (SENS_ON(m)) { SLOOP_BEGIN(is) VFOR(r) SV(m, SV_Y, is, r) = FMA(hg, SV(m, SV_ZN0 + 1, is, r), SV(m, SV_ZN0, is, r)); VEND SLOOP_END }
#endif
    if (BWD) { if (interp_y(m, m.tn + hg) != CV_SUCCESS) { m.nfe++; return CV_RHSFUNC_FAIL; } }
    int retval = cv_f(m, m.tn + hg, m.y, m.tempv);
    if (retval < 0) return CV_RHSFUNC_FAIL;
    if (retval > 0) return RHSFUNC_RECVR;
#ifdef SA_SENS
    if (SENS_ON(m)) {
        retval = cv_fS<SV_Y, SV_TEMPV>(m, m.tn + hg, m.y);
        if (retval < 0) return CV_SRHSFUNC_FAIL;
        if (retval > 0) return SRHSFUNC_RECVR;
    }
#endif
    if (BWD) {
        retval = cv_fQ(m, m.tn + hg, m.y, m.tempvQ);
        if (retval < 0) return CV_QRHSFUNC_FAIL;
        if (retval > 0) return QRHSFUNC_RECVR;
    }
    VFOR(r) {
        m.tempv[r] = m.tempv[r] - m.zn[1][r];
        m.tempv[r] = (1.0 / hg) * m.tempv[r];
    } VEND
    *yddnrm = wrms_n(m, m.tempv, m.ewt);
#ifdef SA_SENS
    if (SENS_ON(m)) {
        SLOOP_BEGIN(is)
            VFOR(r) {
                const double v = SV(m, SV_TEMPV, is, r) - SV(m, SV_ZN0 + 1, is, r);
                SV(m, SV_TEMPV, is, r) = (1.0 / hg) * v;
            } VEND
        SLOOP_END
        *yddnrm = sens_update_norm<SV_TEMPV, SV_EWT>(m, *yddnrm);
    }
#endif
    if (BWD) {
        QFOR(r) {
            m.tempvQ[r] = m.tempvQ[r] - m.znQ[1][r];
            m.tempvQ[r] = (1.0 / hg) * m.tempvQ[r];
        } QEND
        *yddnrm = quad_update_norm(m, *yddnrm, m.tempvQ);
    }
    return CV_SUCCESS;
}

template <bool BWD>
DEV int cv_hin(SA_STATE<BWD> &m, double tout)
{
    double tdiff = tout - m.tn;
    if (tdiff == 0.0) return CV_TOO_CLOSE;
    double sign = (tdiff > 0.0) ? 1.0 : -1.0;
    double tdist = fabs(tdiff);
    double tround = UROUND * fmax(fabs(m.tn), fabs(tout));
    if (tdist < 2.0 * tround) return CV_TOO_CLOSE;
    double hlb = HLB_FACTOR * tround;
    double hub = cv_upper_bound_h0(m, tdist);
    double hg = sqrt(hlb * hub);
    if (hub < hlb) {
        m.h = (sign < 0.0) ? -hg : hg;
        return CV_SUCCESS;
    }
    double hs = hg, hnew = hg, yddnrm = 0.0;
    int result = 1;
    for (int count1 = 1; count1 <= HIN_MAX_ITERS && result == 1; count1++) {
        int hgOK = 0;
        for (int count2 = 1; count2 <= HIN_MAX_ITERS; count2++) {
            double hgs = hg * sign;
            int retval = cv_ydd_norm(m, hgs, &yddnrm);
            if (retval < 0) { result = CV_RHSFUNC_FAIL; break; }
            if (retval == CV_SUCCESS) { hgOK = 1; break; }
            hg *= 0.2;
        }
        if (result != 1) break;
        if (!hgOK) {
            if (count1 <= 2) { result = CV_REPTD_RHSFUNC_ERR; break; }
            hnew = hs;
            result = 0;
            break;
        }
        hs = hg;
        hnew = (yddnrm * hub * hub > 2.0) ? sqrt(2.0 / yddnrm) : sqrt(hg * hub);
        if (count1 == HIN_MAX_ITERS) { result = 0; break; }
        double hrat = hnew / hg;
        if ((hrat > 0.5) && (hrat < 2.0)) { result = 0; break; }
        if ((count1 > 1) && (hrat > 2.0)) { hnew = hg; result = 0; break; }
        hg = hnew;
    }
    if (result < 0) return result;
    double h0 = H_BIAS * hnew;
    if (h0 < hlb) h0 = hlb;
    if (h0 > hub) h0 = hub;
    if (sign < 0.0) h0 = -h0;
    m.h = h0;
    return CV_SUCCESS;
}

/* ---- Nordsieck array manipulation (columns j > q are kept at zero, see bdf_kernels.hip) ---- */
template <bool BWD>
DEV void cv_rescale(SA_STATE<BWD> &m)
{
    double factor = m.eta;
    SFOR(j, 1, (QMAX) + 1) {
        VFOR(r) m.zn[j][r] *= factor; VEND
        if (BWD) { QFOR(r) m.znQ[j][r] *= factor; QEND }
#ifdef SA_SENS
        if (SENS_ON(m)) { SLOOP_BEGIN(is) VFOR(r) SV(m, SV_ZN0 + j, is, r) *= factor; VEND SLOOP_END }
#endif
        factor *= m.eta;
    } SEND
    m.h = m.hscale * m.eta;
    m.hscale = m.h;
}

template <bool BWD>
DEV void cv_increase_bdf(SA_STATE<BWD> &m)
{
    SFOR(i, 0, (QMAX) + 1) m.l[i] = 0.0; SEND
    double alpha1 = 1.0, prod = 1.0, xiold = 1.0, alpha0 = -1.0, hsum = m.hscale;
    m.l[2] = 1.0;
    SFOR(j, 1, QMAX - 1) {
        if (j < m.q) {
            hsum += m.tau[j + 1];
            double xi = hsum / m.hscale;
            prod *= xi;
            alpha0 -= 1.0 / (j + 1);
            alpha1 += 1.0 / xi;
            SFOR_DOWN(i, j + 2, 2) m.l[i] = FMA(m.l[i], xiold, m.l[i - 1]); SEND
            xiold = xi;
        }
    } SEND
    const double A1 = (-alpha0 - alpha1) / prod;
    const int L = m.L;
    TMPV(m, znL, 4); TMPQ(m, znQL, 0);
    VFOR(r) znL[r] = A1 * m.zsave[r]; VEND
    QFOR(r) znQL[r] = BWD ? A1 * m.zsaveQ[r] : 0.0; QEND
    SFOR(j, 2, (QMAX) + 1) {
        if (j == L) {
            VFOR(r) m.zn[j][r] = znL[r]; VEND
            if (BWD) { QFOR(r) m.znQ[j][r] = znQL[r]; QEND }
        }
    } SEND
    SFOR(j, 2, QMAX) {
        if (j <= m.q) {
            VFOR(r) m.zn[j][r] = FMA(m.l[j], znL[r], m.zn[j][r]); VEND
            if (BWD) { QFOR(r) m.znQ[j][r] = FMA(m.l[j], znQL[r], m.znQ[j][r]); QEND }
        }
    } SEND
#ifdef SA_SENS
    if (SENS_ON(m)) {
        SLOOP_BEGIN(is)
            TMPV(m, zl, 5);
            VFOR(r) zl[r] = A1 * SV(m, SV_ZSAVE, is, r); VEND
            SFOR(j, 2, (QMAX) + 1) { if (j == L) { VFOR(r) SV(m, SV_ZN0 + j, is, r) = zl[r]; VEND } } SEND
            SFOR(j, 2, QMAX) {
                if (j <= m.q) { VFOR(r) SV(m, SV_ZN0 + j, is, r) = FMA(m.l[j], zl[r], SV(m, SV_ZN0 + j, is, r)); VEND }
            } SEND
        SLOOP_END
    }
#endif
}

template <bool BWD>
DEV void cv_decrease_bdf(SA_STATE<BWD> &m)
{
    SFOR(i, 0, (QMAX) + 1) m.l[i] = 0.0; SEND
    m.l[2] = 1.0;
    double hsum = 0.0;
    SFOR(j, 1, (QMAX - 2) + 1) {
        if (j <= m.q - 2) {
            hsum += m.tau[j];
            double xi = hsum / m.hscale;
            SFOR_DOWN(i, j + 2, 2) m.l[i] = FMA(m.l[i], xi, m.l[i - 1]); SEND
        }
    } SEND
    TMPV(m, znq, 6); TMPQ(m, znQq, 1);
    VFOR(r) { znq[r] = m.zn[0][r]; SFOR(j, 1, (QMAX) + 1) znq[r] = (m.q == j) ? m.zn[j][r] : znq[r]; SEND } VEND
    QFOR(r) { znQq[r] = m.znQ[0][r]; SFOR(j, 1, (QMAX) + 1) znQq[r] = (m.q == j) ? m.znQ[j][r] : znQq[r]; SEND } QEND
    SFOR(j, 2, QMAX) {
        if (j < m.q) {
            VFOR(r) m.zn[j][r] = FMA(-m.l[j], znq[r], m.zn[j][r]); VEND
            if (BWD) { QFOR(r) m.znQ[j][r] = FMA(-m.l[j], znQq[r], m.znQ[j][r]); QEND }
        }
    } SEND
#ifdef SA_SENS
    if (SENS_ON(m)) {
        SLOOP_BEGIN(is)
            TMPV(m, zq, 7);
            VFOR(r) {
                zq[r] = SV(m, SV_ZN0 + 2, is, r);
                SFOR(k, 3, (QMAX) + 1) { if (m.q == k) zq[r] = SV(m, SV_ZN0 + k, is, r); } SEND
            } VEND
            SFOR(j, 2, QMAX) {
                if (j < m.q) { VFOR(r) SV(m, SV_ZN0 + j, is, r) = FMA(-m.l[j], zq[r], SV(m, SV_ZN0 + j, is, r)); VEND }
            } SEND
        SLOOP_END
    }
#endif
}

template <bool BWD>
DEV void cv_clear_column(SA_STATE<BWD> &m, int q_old)
{
    SFOR(j, 2, (QMAX) + 1) {
        if (j == q_old) {
            VFOR(r) m.zn[j][r] = 0.0; VEND
            if (BWD) { QFOR(r) m.znQ[j][r] = 0.0; QEND }
#ifdef SA_SENS
            if (SENS_ON(m)) { SLOOP_BEGIN(is) VFOR(r) SV(m, SV_ZN0 + j, is, r) = 0.0; VEND SLOOP_END }
#endif
        }
    } SEND
}

template <bool BWD>
DEV void cv_adjust_order(SA_STATE<BWD> &m, int deltaq)
{
    if ((m.q == 2) && (deltaq != 1)) return;
    if (deltaq == 1) cv_increase_bdf(m);
    else if (deltaq == -1) cv_decrease_bdf(m);
}

template <bool BWD>
DEV void cv_predict(SA_STATE<BWD> &m)
{
    m.tn += m.h;
    if (BWD) {
        if ((m.tn - m.tstop) * m.h > 0.0) m.tn = m.tstop;
    }
    SFOR(k, 1, (QMAX) + 1) {
        SFOR_DOWN(j, QMAX, k) {
            VFOR(r) m.zn[j - 1][r] = m.zn[j - 1][r] + m.zn[j][r]; VEND
            if (BWD) { QFOR(r) m.znQ[j - 1][r] = m.znQ[j - 1][r] + m.znQ[j][r]; QEND }
        } SEND
    } SEND
#ifdef SA_SENS
    if (SENS_ON(m)) {           /* the same Pascal-triangle pass, one load and one store per entry */
        SLOOP_BEGIN(is)
            TMPV2(m, z, (QMAX) + 1, 8);
            SFOR(j, 0, (QMAX) + 1) { VFOR(r) z[j][r] = SV(m, SV_ZN0 + j, is, r); VEND } SEND
            SFOR(k, 1, (QMAX) + 1) { SFOR_DOWN(j, QMAX, k) { VFOR(r) z[j - 1][r] = z[j - 1][r] + z[j][r]; VEND } SEND } SEND
            SFOR(j, 0, QMAX) { VFOR(r) SV(m, SV_ZN0 + j, is, r) = z[j][r]; VEND } SEND
        SLOOP_END
    }
#endif
}

template <bool BWD>
DEV void cv_restore(SA_STATE<BWD> &m, double saved_t)
{
    m.tn = saved_t;
    SFOR(k, 1, (QMAX) + 1) {
        SFOR_DOWN(j, QMAX, k) {
            VFOR(r) m.zn[j - 1][r] = m.zn[j - 1][r] - m.zn[j][r]; VEND
            if (BWD) { QFOR(r) m.znQ[j - 1][r] = m.znQ[j - 1][r] - m.znQ[j][r]; QEND }
        } SEND
    } SEND
#ifdef SA_SENS
    if (SENS_ON(m)) {
        SLOOP_BEGIN(is)
            TMPV2(m, z, (QMAX) + 1, 8);
            SFOR(j, 0, (QMAX) + 1) { VFOR(r) z[j][r] = SV(m, SV_ZN0 + j, is, r); VEND } SEND
            SFOR(k, 1, (QMAX) + 1) { SFOR_DOWN(j, QMAX, k) { VFOR(r) z[j - 1][r] = z[j - 1][r] - z[j][r]; VEND } SEND } SEND
            SFOR(j, 0, QMAX) { VFOR(r) SV(m, SV_ZN0 + j, is, r) = z[j][r]; VEND } SEND
        SLOOP_END
    }
#endif
}

template <bool BWD>
DEV int cv_nls_lsetup(SA_STATE<BWD> &m, int jbad, int &convfail)
{
    if (jbad) convfail = CV_FAIL_BAD_J;
    int retval = cv_lsetup(m, convfail);
    m.nsetups++;
    m.nls_jcur = m.jcur;
    m.gamrat = 1.0;
    m.gammap = m.gamma;
    m.crate = 1.0;
#ifdef SA_SENS
    m.crateS = 1.0;
#endif
    m.nstlp = m.nst;
    if (retval < 0) return CV_LSETUP_FAIL;
    if (retval > 0) return NLS_CONV_RECVR;
    return CV_SUCCESS;
}

template <bool BWD, class VR>
DEV int cv_nls_residual(SA_STATE<BWD> &m, VR &res)
{
    VFOR(r) m.y[r] = m.zn[0][r] + m.acor[r]; VEND
    int retval = cv_f(m, m.tn, m.y, m.ftemp);
    if (retval < 0) return CV_RHSFUNC_FAIL;
    if (retval > 0) return RHSFUNC_RECVR;
    VFOR(r) {
        res[r] = FMA(m.rl1, m.zn[1][r], m.acor[r]);
        res[r] = FMA(-m.gamma, m.ftemp[r], res[r]);
    } VEND
    return CV_SUCCESS;
}

template <bool BWD>
DEV int cv_newton_pass(SA_STATE<BWD> &m, int callSetup, int jbad, int &convfail, int &in_loop)
{
    TMPV(m, delta, 14);
#ifdef SA_SENS
    const bool sim = SENS_ON(m) && m.ism == 0;
#endif
    in_loop = 0;
    VFOR(r) m.acor[r] = 0.0; VEND
#ifdef SA_SENS
    if (sim) { SLOOP_BEGIN(is) VFOR(r) SV(m, SV_ACOR, is, r) = 0.0; VEND SLOOP_END }
#endif
    int retval = cv_nls_residual(m, delta);
    if (retval != CV_SUCCESS) return retval;
#ifdef SA_SENS
    if (sim) {
        retval = cv_nls_residual_sens(m);
        if (retval != CV_SUCCESS) return retval;
    }
#endif
    if (callSetup) {
        retval = cv_nls_lsetup(m, jbad, convfail);
        if (retval != CV_SUCCESS) return retval;
    }
    int curiter = 0;
    in_loop = 1;
    for (;;) {
        m.nni++;
        VFOR(r) delta[r] = -1.0 * delta[r]; VEND
        dense_getrs(m, delta);
#ifdef SA_SENS
        if (m.gamrat != 1.0) {
            double s = 2.0 / (1.0 + m.gamrat);
            VFOR(r) delta[r] *= s; VEND
        }
#else
        {   /* cvLsSolve's scaling by 2 / (1 + gamrat) -- skipped by CVODES when gamrat == 1, where the factor is exactly
               1.0: computed by every lane (x * 1.0 == x) instead of inside a divergent block of its own */
            const double s = 2.0 / (1.0 + m.gamrat);
            VFOR(r) delta[r] *= s; VEND
        }
#endif
        VFOR(r) m.acor[r] = m.acor[r] + delta[r]; VEND
        double del = wrms_n(m, delta, m.ewt);
#ifdef SA_SENS
        if (sim) {
            cv_sens_newton_update(m);
            del = sens_update_norm<SV_DELTA, SV_EWT>(m, del);
        }
#endif
        if (curiter > 0) m.crate = fmax(CRDOWN * m.crate, del / m.delp);
        double dcon = del * fmin(1.0, m.crate) * m.tq[4];
        if (dcon <= 1.0) {
            m.acnrm = (curiter == 0) ? del : wrms_n(m, m.acor, m.ewt);
#ifdef SA_SENS
            if (sim && curiter != 0) m.acnrm = sens_update_norm<SV_ACOR, SV_EWT>(m, m.acnrm);
#endif
            m.nls_jcur = 0;
            return CV_SUCCESS;
        }
        if ((curiter >= 1) && (del > RDIV * m.delp)) return NLS_CONV_RECVR;
        m.delp = del;
        curiter++;
        if (curiter >= NLS_MAXCOR) return NLS_CONV_RECVR;
        retval = cv_nls_residual(m, delta);
        if (retval != CV_SUCCESS) return retval;
#ifdef SA_SENS
        if (sim) {
            retval = cv_nls_residual_sens(m);
            if (retval != CV_SUCCESS) return retval;
        }
#endif
    }
}

#ifdef SA_SENS
/* cvStgrNls (ism = CV_STAGGERED): Newton on the sensitivity systems with the state fixed */
template <bool BWD>
DEV int cv_stgr_nls(SA_STATE<BWD> &m)
{
    int callSetup = 0, jbad = 0, convfail = CV_FAIL_OTHER, retval;
    SLOOP_BEGIN(is) VFOR(r) SV(m, SV_ACOR, is, r) = 0.0; VEND SLOOP_END
    for (;;) {
        retval = cv_nls_residual_sens(m);
        if (retval != CV_SUCCESS) break;
        if (callSetup) {
            retval = cv_nls_lsetup(m, jbad, convfail);
            m.nsetupsS++;
            if (retval != CV_SUCCESS) break;
        }
        int curiter = 0;
        for (;;) {
            m.nniS++;
            cv_sens_newton_update(m);
            double del = sens_update_norm<SV_DELTA, SV_EWT>(m, 0.0);
            if (curiter > 0) m.crateS = fmax(CRDOWN * m.crateS, del / m.delpS);
            double dcon = del * fmin(1.0, m.crateS) * m.tq[4];
            if (dcon <= 1.0) {
                m.acnrmS = (curiter == 0) ? del : sens_update_norm<SV_ACOR, SV_EWT>(m, 0.0);
                retval = CV_SUCCESS;
                m.nls_jcur = 0;
                break;
            }
            if ((curiter >= 1) && (del > RDIV * m.delpS)) { retval = NLS_CONV_RECVR; break; }
            m.delpS = del;
            curiter++;
            if (curiter >= NLS_MAXCOR) { retval = NLS_CONV_RECVR; break; }
            retval = cv_nls_residual_sens(m);
            if (retval != CV_SUCCESS) break;
        }
        if (retval == CV_SUCCESS) break;
        if ((retval > 0) && !m.nls_jcur) {
            callSetup = 1;
            jbad = 1;
            SLOOP_BEGIN(is) VFOR(r) SV(m, SV_ACOR, is, r) = 0.0; VEND SLOOP_END
            continue;
        }
        break;
    }
    if (retval != CV_SUCCESS) return retval;
    SLOOP_BEGIN(is) VFOR(r) SV(m, SV_Y, is, r) = SV(m, SV_ZN0, is, r) + SV(m, SV_ACOR, is, r); VEND SLOOP_END
    return CV_SUCCESS;
}
#endif

template <bool BWD>
DEV int cv_error_test_failed(SA_STATE<BWD> &m, double saved_t, double dsm, int &nef, int &netf_counter)
{
    nef++;
    netf_counter++;
    cv_restore(m, saved_t);
    if (nef == MXNEF) return CV_ERR_FAILURE;
    m.etamax = 1.0;
    if (nef <= MXNEF1) {
        m.eta = 1.0 / (rpower_r<SA_POLY_CM(BWD)>(BIAS2 * dsm, inv_int(m.L)) + ADDON);
        m.eta = fmax(ETAMIN, m.eta);
        if (nef >= SMALL_NEF) m.eta = fmin(m.eta, ETAMXF);
        cv_rescale(m);
        return 0;
    }
    if (m.q > 1) {
        m.eta = ETAMIN;
        cv_adjust_order(m, -1);
        cv_clear_column(m, m.q);
        m.L = m.q;
        m.q--;
        m.qwait = m.L;
        cv_rescale(m);
        return 0;
    }
    m.eta = ETAMIN;
    m.h *= m.eta;
    m.hscale = m.h;
    m.qwait = LONG_WAIT;
    if (BWD) { if (interp_y(m, m.tn) != CV_SUCCESS) { m.nfe++; return CV_RHSFUNC_FAIL; } }
    int retval = cv_f(m, m.tn, m.zn[0], m.tempv);
    if (retval < 0) return CV_RHSFUNC_FAIL;
    if (retval > 0) return CV_UNREC_RHSFUNC_ERR;
    VFOR(r) m.zn[1][r] = m.h * m.tempv[r]; VEND
#ifdef SA_SENS
    if (SENS_ON(m)) {
        retval = cv_fS<SV_ZN0, SV_TEMPV>(m, m.tn, m.zn[0]);
        if (retval < 0) return CV_SRHSFUNC_FAIL;
        if (retval > 0) return CV_UNREC_SRHSFUNC_ERR;
        SLOOP_BEGIN(is) VFOR(r) SV(m, SV_ZN0 + 1, is, r) = m.h * SV(m, SV_TEMPV, is, r); VEND SLOOP_END
    }
#endif
    if (BWD) {
        retval = cv_fQ(m, m.tn, m.zn[0], m.tempvQ);
        if (retval < 0) return CV_QRHSFUNC_FAIL;
        if (retval > 0) return CV_UNREC_QRHSFUNC_ERR;
        QFOR(r) m.znQ[1][r] = m.h * m.tempvQ[r]; QEND
    }
    return 0;
}

template <bool BWD>
DEV void cv_complete_step(SA_STATE<BWD> &m)
{
    m.nst++;
    m.hu = m.h;
    m.qu = m.q;
    SFOR_DOWN(i, QMAX, 2) m.tau[i] = (i <= m.q) ? m.tau[i - 1] : m.tau[i]; SEND
    m.tau[2] = ((m.q == 1) && (m.nst > 1)) ? m.tau[1] : m.tau[2];
    m.tau[1] = m.h;
    SFOR(j, 0, (QMAX) + 1) {
        VFOR(r) m.zn[j][r] = FMA(m.l[j], m.acor[r], m.zn[j][r]); VEND
        if (BWD) { QFOR(r) m.znQ[j][r] = FMA(m.l[j], m.acorQ[r], m.znQ[j][r]); QEND }
    } SEND
#ifdef SA_SENS
    if (SENS_ON(m)) {
        SLOOP_BEGIN(is)
            TMPV(m, ac, 15);
            VFOR(r) ac[r] = SV(m, SV_ACOR, is, r); VEND
            SFOR(j, 0, (QMAX) + 1) { VFOR(r) SV(m, SV_ZN0 + j, is, r) = FMA(m.l[j], ac[r], SV(m, SV_ZN0 + j, is, r)); VEND } SEND
            if ((m.qwait - 1 == 1) && (m.q != QMAX)) { VFOR(r) SV(m, SV_ZSAVE, is, r) = ac[r]; VEND }
        SLOOP_END
    }
#endif
    m.qwait--;
    {
        const bool sv = (m.qwait == 1) && (m.q != QMAX);
        VFOR(r) m.zsave[r] = sv ? m.acor[r] : m.zsave[r]; VEND
        if (BWD) { QFOR(r) m.zsaveQ[r] = sv ? m.acorQ[r] : m.zsaveQ[r]; QEND }
        m.saved_tq5 = sv ? m.tq[5] : m.saved_tq5;
    }
}

template <bool BWD>
DEV void cv_set_eta(SA_STATE<BWD> &m)
{
    if (m.eta < THRESH) {
        m.eta = 1.0;
        m.hprime = m.h;
    } else {
        m.eta = fmin(m.eta, m.etamax);
        m.hprime = m.h * m.eta;
    }
}

template <bool BWD>
DEV void cv_prepare_next_step(SA_STATE<BWD> &m, double dsm)
{
    if (m.etamax == 1.0) {
        m.qwait = m.qwait > 2 ? m.qwait : 2;
        m.qprime = m.q;
        m.hprime = m.h;
        m.eta = 1.0;
        return;
    }
    /* cvComputeEtaqm1 / cvComputeEtaqp1 / cvChooseEta as ONE straight-line block (see bdf_kernels.hip): with 64/G
       instances per wavefront some group is at an order decision in nearly every iteration, so the full path runs
       anyway; here its two norms and three powers are independent chains of one basic block, groups that are not at
       a decision (qwait != 0) or whose candidate is not defined discard the values through selects.  Values and
       written fields identical to the branching form. */
    const bool full = (m.qwait == 0);
    if (!SA_SHORTCUT(full)) {
        /* no lane of the wavefront is at an order decision (after a restart the lanes walk through qwait together):
           what the block below leaves behind for full == false, without its two norms and two of its three powers */
        const double p0 = rpower_nb<SA_POLY_CM(BWD)>(BIAS2 * dsm, inv_int(m.L));
        const double etaq = 1.0 / (p0 + ADDON);
        m.etaq = etaq;
        m.qprime = m.q;
        const bool small = etaq < THRESH;
        const double capped = fmin(etaq, m.etamax);
        m.hprime = small ? m.h : m.h * capped;
        m.eta = small ? 1.0 : capped;
        return;
    }
    TMPV(m, znq, 16); TMPQ(m, znQq, 2); TMPV(m, tv, 17); TMPQ(m, tvQ, 3);
    VFOR(r) { znq[r] = m.zn[0][r]; SFOR(j, 1, (QMAX) + 1) znq[r] = (m.q == j) ? m.zn[j][r] : znq[r]; SEND } VEND
    QFOR(r) { znQq[r] = m.znQ[0][r]; SFOR(j, 1, (QMAX) + 1) znQq[r] = (m.q == j) ? m.znQ[j][r] : znQq[r]; SEND } QEND
    double ddn = wrms_n(m, znq, m.ewt);
    if (BWD) ddn = quad_update_norm(m, ddn, znQq);
#ifdef SA_SENS
    if (SENS_ON(m) && full && m.q > 1) {            /* cvComputeEtaqm1: the sensitivities' column q takes part */
        SLOOP_BEGIN(is)
            VFOR(r) {
                double v = SV(m, SV_ZN0 + 2, is, r);
                SFOR(k, 3, (QMAX) + 1) { if (m.q == k) v = SV(m, SV_ZN0 + k, is, r); } SEND
                SV(m, SV_TEMPV, is, r) = v;
            } VEND
        SLOOP_END
        ddn = sens_update_norm<SV_TEMPV, SV_EWT>(m, ddn);
    }
#endif
    ddn = ddn * m.tq[1];
    const double base = m.h / m.tau[2];
    double pw = 1.0;
    SFOR(i, 1, (QMAX + 1) + 1) { pw = (i <= m.L) ? pw * base : pw; } SEND
    const double cquot = (m.tq[5] / m.saved_tq5) * pw;
    VFOR(r) tv[r] = FMA(-cquot, m.zsave[r], m.acor[r]); VEND
    double dup = wrms_n(m, tv, m.ewt);
    if (BWD) {
        QFOR(r) tvQ[r] = FMA(-cquot, m.zsaveQ[r], m.acorQ[r]); QEND
        dup = quad_update_norm(m, dup, tvQ);
    }
#ifdef SA_SENS
    if (SENS_ON(m) && full && (m.q != QMAX) && (m.saved_tq5 != 0.0)) {     /* cvComputeEtaqp1 */
        SLOOP_BEGIN(is) VFOR(r) SV(m, SV_TEMPV, is, r) = FMA(-cquot, SV(m, SV_ZSAVE, is, r), SV(m, SV_ACOR, is, r)); VEND SLOOP_END
        dup = sens_update_norm<SV_TEMPV, SV_EWT>(m, dup);
    }
#endif
    dup = dup * m.tq[3];
    const double p0 = rpower_nb<SA_POLY_CM(BWD)>(BIAS2 * dsm, inv_int(m.L));
    const double p1 = rpower_nb<SA_POLY_CM(BWD)>(BIAS1 * ddn, inv_int(m.q));
    const double p2 = rpower_nb<SA_POLY_CM(BWD)>(BIAS3 * dup, inv_int(m.L + 1));
    const double etaq = 1.0 / (p0 + ADDON), e1 = 1.0 / (p1 + ADDON), e2 = 1.0 / (p2 + ADDON);
    const double etaqm1 = (m.q > 1) ? e1 : 0.0;
    const double etaqp1 = ((m.q != QMAX) && (m.saved_tq5 != 0.0)) ? e2 : 0.0;
    m.etaq = etaq;
    m.etaqm1 = full ? etaqm1 : m.etaqm1;
    m.etaqp1 = full ? etaqp1 : m.etaqp1;
    m.qwait = full ? 2 : m.qwait;
    const double etam = fmax(etaqm1, fmax(etaq, etaqp1));
    const bool c0 = etam < THRESH, c1 = (etam == etaq), c2 = (etam == etaqm1);
    const double eta_f = c0 ? 1.0 : (c1 ? etaq : (c2 ? etaqm1 : etaqp1));
    const int qp_f = c0 ? m.q : (c1 ? m.q : (c2 ? m.q - 1 : m.q + 1));
    const bool up = full && !c0 && !c1 && !c2;
    m.eta = full ? eta_f : etaq;
    m.qprime = full ? qp_f : m.q;
    VFOR(r) m.zsave[r] = up ? m.acor[r] : m.zsave[r]; VEND
    if (BWD) { QFOR(r) m.zsaveQ[r] = up ? m.acorQ[r] : m.zsaveQ[r]; QEND }
#ifdef SA_SENS
    if (SENS_ON(m) && up) { SLOOP_BEGIN(is) VFOR(r) SV(m, SV_ZSAVE, is, r) = SV(m, SV_ACOR, is, r); VEND SLOOP_END }
#endif
    {   /* cvSetEta */
        const bool small = m.eta < THRESH;
        const double capped = fmin(m.eta, m.etamax);
        m.hprime = small ? m.h : m.h * capped;
        m.eta = small ? 1.0 : capped;
    }
}

template <bool BWD, class VD, class VDQ>
DEV int cv_get_dky0(const SA_STATE<BWD> &m, double t, VD &&dky, VDQ &&dkyQ)
{
    double tfuzz = FUZZ_FACTOR * UROUND * (fabs(m.tn) + fabs(m.hu));
    if (m.hu < 0.0) tfuzz = -tfuzz;
    double tp = m.tn - m.hu - tfuzz;
    double tn1 = m.tn + tfuzz;
    if ((t - tp) * (t - tn1) > 0.0) return CV_BAD_T;
    double s = (t - m.tn) / m.h;
    double pw[QMAX + 1];
    pw[0] = 1.0;
    SFOR(j, 1, (QMAX) + 1) pw[j] = pw[j - 1] * s; SEND
    VFOR(r) {
        double acc = pw[QMAX] * m.zn[QMAX][r];
        SFOR_DOWN(j, QMAX - 1, 0) acc = FMA(pw[j], m.zn[j][r], acc); SEND
        dky[r] = acc;
    } VEND
    if (BWD) {
        QFOR(r) {
            double acc = pw[QMAX] * m.znQ[QMAX][r];
            SFOR_DOWN(j, QMAX - 1, 0) acc = FMA(pw[j], m.znQ[j][r], acc); SEND
            dkyQ[r] = acc;
        } QEND
    }
    return CV_SUCCESS;
}

template <bool BWD>
DEV int cv_first_call(SA_STATE<BWD> &m, double tout)
{
#ifdef SA_CONSTRAINTS
    if (!BWD && m.constr) {
        double bad = 0.0;
        VFOR(r) bad = ((IDX(m, r) < NS) && constr_violated(m.cons[r], m.zn[0][r])) ? 1.0 : bad; VEND
        if (wave_max(m.lane, bad) > 0.0) return CV_ILL_INPUT;
    }
#endif
    if (ewt_set(m, m.zn[0], m.ewt) != 0) return CV_ILL_INPUT;
    if (BWD) { if (ewtQ_set(m, m.znQ[0], m.ewtQ) != 0) return CV_ILL_INPUT; }
#ifdef SA_SENS
    if (SENS_ON(m)) { if (sens_ewt_set<SV_ZN0, SV_EWT>(m) != 0) return CV_ILL_INPUT; }
#endif
    if (BWD) { if (interp_y(m, m.tn) != CV_SUCCESS) { m.nfe++; return CV_RHSFUNC_FAIL; } }
    int retval = cv_f(m, m.tn, m.zn[0], m.zn[1]);
    if (retval < 0) return CV_RHSFUNC_FAIL;
    if (retval > 0) return CV_FIRST_RHSFUNC_ERR;
#ifdef SA_HERMITE
    VFOR(r) m.f0[r] = m.zn[1][r]; VEND
#endif
    if (BWD) {
        retval = cv_fQ(m, m.tn, m.zn[0], m.znQ[1]);
        if (retval < 0) return CV_QRHSFUNC_FAIL;
        if (retval > 0) return CV_FIRST_QRHSFUNC_ERR;
    }
#ifdef SA_SENS
    if (SENS_ON(m)) {
        retval = cv_fS<SV_ZN0, SV_ZN0 + 1>(m, m.tn, m.zn[0]);
        if (retval < 0) return CV_SRHSFUNC_FAIL;
        if (retval > 0) return CV_FIRST_SRHSFUNC_ERR;
    }
#endif
    double tout_hin = tout;
    if (BWD) {
        if ((m.tstop - m.tn) * (tout - m.tn) <= 0.0) return CV_ILL_INPUT;
        if ((tout - m.tn) * (tout - m.tstop) > 0.0) tout_hin = m.tstop;
    }
    int hflag = cv_hin(m, tout_hin);
    if (hflag != CV_SUCCESS) return hflag;
    if (BWD) {
        if ((m.tn + m.h - m.tstop) * m.h > 0.0) m.h = (m.tstop - m.tn) * (1.0 - 4.0 * UROUND);
    }
    m.hscale = m.h;
    m.hprime = m.h;
    VFOR(r) m.zn[1][r] = m.h * m.zn[1][r]; VEND
    if (BWD) { QFOR(r) m.znQ[1][r] = m.h * m.znQ[1][r]; QEND }
#ifdef SA_SENS
    if (SENS_ON(m)) { SLOOP_BEGIN(is) VFOR(r) SV(m, SV_ZN0 + 1, is, r) = m.h * SV(m, SV_ZN0 + 1, is, r); VEND SLOOP_END }
#endif
    return CV_SUCCESS;
}

template <bool BWD>
DEV int cv_pre_step(SA_STATE<BWD> &m)
{
#if defined(SA_PRESTEP_FUSED) && !defined(SA_SENS)
    /* Mappings that can hand over the MEAN SQUARE of a weighted vector (the argument of the WRMS norm's square root):
       both weight vectors and the CV_TOO_MUCH_ACC test in one straight line -- no early returns (each one is a
       divergent region of its own), no square roots.  UROUND * sqrt(x) > 1 with UROUND = 2^-52 and a correctly
       rounded, monotone sqrt holds exactly for x >= 2^104 + 2^53 (the first double whose root rounds above 2^52), and
       max(sqrt(a), sqrt(b)) exceeds the bound iff max(a, b) does; NaNs fail every comparison in both forms. */
    int bad = ewt_set(m, m.zn[0], m.ewt);
    double ms = wrms2_n(m, m.zn[0], m.ewt);
    if (BWD) {
        bad |= ewtQ_set(m, m.znQ[0], m.ewtQ);
        const double mq = wrms2_q(m, m.znQ[0], m.ewtQ);
        ms = ms > mq ? ms : mq;
    }
    const bool acc = ms >= 0x1.0000000000002p+104;
    return bad ? CV_ILL_INPUT : (acc ? CV_TOO_MUCH_ACC : CV_SUCCESS);
#endif
    if (ewt_set(m, m.zn[0], m.ewt) != 0) return CV_ILL_INPUT;
    if (BWD) { if (ewtQ_set(m, m.znQ[0], m.ewtQ) != 0) return CV_ILL_INPUT; }
#ifdef SA_SENS
    if (SENS_ON(m)) { if (sens_ewt_set<SV_ZN0, SV_EWT>(m) != 0) return CV_ILL_INPUT; }
#endif
    double nrm = wrms_n(m, m.zn[0], m.ewt);
    if (BWD) nrm = quad_update_norm(m, nrm, m.znQ[0]);
#ifdef SA_SENS
    if (SENS_ON(m)) nrm = sens_update_norm<SV_ZN0, SV_EWT>(m, nrm);
#endif
    if (UROUND * nrm > 1.0) return CV_TOO_MUCH_ACC;
    return CV_SUCCESS;
}

struct StepCtl {
    int in_step, redo, nflag, ncf, nef, nefQ, convfail, ncfS, nefS;
    double saved_t;
};



template <bool BWD>
DEV int cv_handle_nflag_failed(SA_STATE<BWD> &m, StepCtl &c, int nflag, int &ncf, int &ncfn)
{
    ncfn++;
    cv_restore(m, c.saved_t);
    if (nflag < 0) return nflag;
    ncf++;
    m.etamax = 1.0;
    if (ncf == MXNCF) {
        if (nflag == NLS_CONV_RECVR) return CV_CONV_FAILURE;
        if (nflag == RHSFUNC_RECVR) return CV_REPTD_RHSFUNC_ERR;
        if (nflag == SRHSFUNC_RECVR) return CV_REPTD_SRHSFUNC_ERR;
        if (nflag == CONSTR_RECVR) return CV_CONSTR_FAIL;
        return CV_REPTD_QRHSFUNC_ERR;
    }
    if (nflag != CONSTR_RECVR) m.eta = ETACF;         /* CONSTR_RECVR: eta was set by the constraint check */
    c.nflag = PREV_CONV_FAIL;
    cv_rescale(m);
    return 0;
}

/* one step ATTEMPT; 1 = step completed, 0 = call again, <0 = unrecoverable (see bdf_kernels.hip) */
template <bool BWD>
DEV int cv_attempt(SA_STATE<BWD> &m, StepCtl &c)
{
    if (!c.in_step) {
        c.saved_t = m.tn;
        c.ncf = c.nef = c.nefQ = 0;
        c.ncfS = c.nefS = 0;
        c.nflag = FIRST_CALL;
        c.redo = 0;
#ifdef SA_RESCALE_ALWAYS
        {   /* cvAdjustParams, one-lane-per-instance form: the (rare) order change stays a branch, the rescale runs for
               every lane -- with eta = 1 (an exact no-op) where the step size does not change -- instead of as a
               divergent block some lane of the wavefront takes in nearly every iteration */
            const bool adj = (m.nst > 0) && (m.hprime != m.h);
            if (adj && (m.qprime != m.q)) {
                cv_adjust_order(m, m.qprime - m.q);
                if (m.qprime < m.q) cv_clear_column(m, m.q);
                m.q = m.qprime;
                m.L = m.q + 1;
                m.qwait = m.L;
            }
            const double eta_keep = m.eta, h_keep = m.h, hs_keep = m.hscale;
            m.eta = adj ? m.eta : 1.0;
            cv_rescale(m);
            m.eta = eta_keep;
            m.h = adj ? m.h : h_keep;
            m.hscale = adj ? m.hscale : hs_keep;
        }
#else
        if ((m.nst > 0) && (m.hprime != m.h)) {
            if (m.qprime != m.q) {
                cv_adjust_order(m, m.qprime - m.q);
                if (m.qprime < m.q) cv_clear_column(m, m.q);
                m.q = m.qprime;
                m.L = m.q + 1;
                m.qwait = m.L;
            }
            cv_rescale(m);
        }
#endif
        c.in_step = 1;
    }
    int callSetup, jbad;
    PH_T0
    if (!c.redo) {
        cv_predict(m);
        cv_set(m);
        PH_ADD(m, 1)
        COLD_STORE(m);          /* until the Newton pass (and the quadrature callback) are over */
        if (BWD) { if (interp_y(m, m.tn) != CV_SUCCESS) { m.nfe++; return CV_RHSFUNC_FAIL; } }
        PH_ADD(m, 2)
        c.convfail = ((c.nflag == FIRST_CALL) || (c.nflag == PREV_ERR_FAIL)) ? CV_NO_FAILURES : CV_FAIL_OTHER;
        callSetup = (c.nflag == PREV_CONV_FAIL) || (c.nflag == PREV_ERR_FAIL) || (m.nst == 0) ||
                    (m.nst >= m.nstlp + MSBP) || (fabs(m.gamrat - 1.0) > DGMAX);
        jbad = 0;
    } else {
        callSetup = 1;
        jbad = 1;
        COLD_STORE(m);
    }
    int in_loop;
    int nls = cv_newton_pass(m, callSetup, jbad, c.convfail, in_loop);
    PH_ADD(m, 3)
    if ((nls > 0) && in_loop && !m.nls_jcur) {
        COLD_LOAD(m);
        c.redo = 1;
        return 0;
    }
    c.redo = 0;
    if (nls != CV_SUCCESS) { COLD_LOAD(m); return cv_handle_nflag_failed(m, c, nls, c.ncf, m.ncfn); }

    VFOR(r) m.y[r] = m.zn[0][r] + m.acor[r]; VEND
#ifdef SA_CONSTRAINTS
    if (!BWD && m.constr) {             /* cvCheckConstraints (see the oracle) */
        TMPV(m, mm, 18); TMPV(m, v, 19);
        double anyv = 0.0;
        VFOR(r) {
            const bool bad = (IDX(m, r) < NS) && constr_violated(m.cons[r], m.y[r]);
            mm[r] = bad ? 1.0 : 0.0;
            anyv = bad ? 1.0 : anyv;
        } VEND
        if (wave_max(m.lane, anyv) > 0.0) {
            VFOR(r) {
                const double aa = (fabs(m.cons[r]) >= 1.5) ? 1.0 : 0.0;
                double tmp = (aa * m.cons[r]) / m.ewt[r];
                tmp = FMA(-0.1, tmp, m.y[r]);
                v[r] = (IDX(m, r) < NS) ? tmp * mm[r] : 0.0;
            } VEND
            const double vnorm = wrms_n(m, v, m.ewt);
            if (vnorm * m.tq[4] <= 1.0) {
                VFOR(r) m.acor[r] = m.acor[r] - v[r]; VEND
            } else {
                double q = 1e308;
                VFOR(r) {
                    const double d = mm[r] * (m.zn[0][r] - m.y[r]);
                    const double qv = (IDX(m, r) < NS && d != 0.0) ? m.zn[0][r] / d : 1e308;
                    q = qv < q ? qv : q;
                } VEND
                const double minq = -wave_max(m.lane, -q);
                m.eta = fmax(0.9 * minq, 0.1);
                COLD_LOAD(m);
                return cv_handle_nflag_failed(m, c, CONSTR_RECVR, c.ncf, m.ncfn);
            }
        }
    }
#endif
    double dsm = m.acnrm * m.tq[2];
    if (dsm > 1.0) {
        c.nflag = PREV_ERR_FAIL;
        COLD_LOAD(m);
        return cv_error_test_failed(m, c.saved_t, dsm, c.nef, m.netf);
    }
#ifdef SA_SENS
    if (SENS_ON(m) && m.ism == 0) { SLOOP_BEGIN(is) VFOR(r) SV(m, SV_Y, is, r) = SV(m, SV_ZN0, is, r) + SV(m, SV_ACOR, is, r); VEND SLOOP_END }
    if (SENS_ON(m) && m.ism == 1) {      /* CV_STAGGERED: sensitivities after the state passed (oracle cv_step) */
        c.ncf = c.nef = 0;
        int retval = cv_f(m, m.tn, m.y, m.ftemp);
        if (retval < 0) return CV_RHSFUNC_FAIL;
        if (retval > 0) { COLD_LOAD(m); c.nflag = PREV_CONV_FAIL; return 0; }
        const int nflagS = cv_stgr_nls(m);
        if (nflagS != CV_SUCCESS) { COLD_LOAD(m); return cv_handle_nflag_failed(m, c, nflagS, c.ncfS, m.ncfnS); }
        m.acnrmS = sens_update_norm<SV_ACOR, SV_EWT>(m, 0.0);
        const double dsmS = m.acnrmS * m.tq[2];
        if (dsmS > 1.0) {
            c.nflag = PREV_ERR_FAIL;
            COLD_LOAD(m);
            return cv_error_test_failed(m, c.saved_t, dsmS, c.nefS, m.netfS);
        }
        if (dsmS > dsm) dsm = dsmS;
    }
#endif
    if (BWD) {
        c.ncf = c.nef = 0;
        int retval = cv_fQ(m, m.tn, m.y, m.acorQ);
        COLD_LOAD(m);
        if (retval != 0) return cv_handle_nflag_failed(m, c, retval < 0 ? CV_QRHSFUNC_FAIL : QRHSFUNC_RECVR, c.ncf, m.ncfn);
        QFOR(r) {
            m.acorQ[r] = FMA(m.h, m.acorQ[r], -m.znQ[1][r]);
            m.acorQ[r] = m.rl1 * m.acorQ[r];
        } QEND
        double acnrmQ = wrms_q(m, m.acorQ, m.ewtQ);
        double dsmQ = acnrmQ * m.tq[2];
        if (dsmQ > 1.0) {
            c.nflag = PREV_ERR_FAIL;
            return cv_error_test_failed(m, c.saved_t, dsmQ, c.nefQ, m.netfQ);
        }
        if (dsmQ > dsm) dsm = dsmQ;
    } else {
        COLD_LOAD(m);
    }
    PH_ADD(m, 4)
    cv_complete_step(m);
    cv_prepare_next_step(m, dsm);
    m.etamax = (m.nst <= SMALL_NST) ? ETAMX2 : ETAMX3;
    VFOR(r) m.acor[r] = m.tq[2] * m.acor[r]; VEND
    if (BWD) { QFOR(r) m.acorQ[r] = m.tq[2] * m.acorQ[r]; QEND }
#ifdef SA_SENS
    if (SENS_ON(m)) { SLOOP_BEGIN(is) VFOR(r) SV(m, SV_ACOR, is, r) = m.tq[2] * SV(m, SV_ACOR, is, r); VEND SLOOP_END }
#endif
    c.in_step = 0;
    PH_ADD(m, 5)
    return 1;
}

template <bool BWD>
DEV void accumulate_stats(const SA_STATE<BWD> &m, int64_t *acc)
{
    acc[ST_NST] += m.nst; acc[ST_NFE] += m.nfe; acc[ST_NSETUPS] += m.nsetups; acc[ST_NJE] += m.nje;
    acc[ST_NNI] += m.nni; acc[ST_NCFN] += m.ncfn; acc[ST_NETF] += m.netf; acc[ST_QLAST] = m.qu;
    acc[ST_NFQE] += m.nfQe; acc[ST_NETFQ] += m.netfQ;
}


#endif
