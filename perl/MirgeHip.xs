/* MirgeHip.xs -- thin XS binding of libmirge_hip.so (include/mirge_hip.h) for the reference's
 * Perl driver.  One XSUB per C-ABI entry point the hot path of miRge.pl needs; INTEGRATION.md shows
 * the few lines that replace quantReads / runAnnotationPipeline / summarize (miRge.pl:291-494).
 * FFI::Platypus is not installed in the image, hence XS (xsubpp + the system perl headers). */
#define PERL_NO_GET_CONTEXT
#include "EXTERN.h"
#include "perl.h"
#include "XSUB.h"
#include "../include/mirge_hip.h"

typedef mirge_ctx* MirgeHip;
typedef mirge_group* MirgeHipGroup;

MODULE = MirgeHip    PACKAGE = MirgeHip    PREFIX = mirge_

PROTOTYPES: DISABLE

MirgeHip
mirge_open(species_dir, n_samples, device = 0)
        const char* species_dir
        int n_samples
        int device
    CODE:
        RETVAL = mirge_open(species_dir, n_samples, device);
        if (!RETVAL) croak("MirgeHip::open: %s", mirge_last_error(NULL));
    OUTPUT:
        RETVAL

void
mirge_close(ctx)
        MirgeHip ctx

const char*
mirge_last_error(ctx)
        MirgeHip ctx

int
mirge_lib_size(ctx, lib)
        MirgeHip ctx
        int lib

const char*
mirge_lib_name(ctx, lib, idx)
        MirgeHip ctx
        int lib
        int idx

const char*
mirge_lib_short_name(ctx, lib, idx)
        MirgeHip ctx
        int lib
        int idx

UV
mirge_collapse_fastq(ctx, path, sample)
        MirgeHip ctx
        const char* path
        int sample
    PREINIT:
        uint64_t n = 0;
    CODE:
        if (mirge_collapse_fastq(ctx, path, sample, &n) != MIRGE_OK) croak("MirgeHip::collapse_fastq: %s", mirge_last_error(ctx));
        RETVAL = (UV)n;
    OUTPUT:
        RETVAL

void
mirge_collapse_raw_fastq(ctx, path, sample, adapter, trim5, trim3)
        MirgeHip ctx
        const char* path
        int sample
        const char* adapter
        int trim5
        int trim3
    PREINIT:
        uint64_t total = 0, kept = 0; int32_t phred = 33;
    PPCODE:
        if (mirge_collapse_raw_fastq(ctx, path, sample, adapter, trim5, trim3, &total, &kept, &phred) != MIRGE_OK)
            croak("MirgeHip::collapse_raw_fastq: %s", mirge_last_error(ctx));
        EXTEND(SP, 3);
        PUSHs(sv_2mortal(newSVuv((UV)total)));
        PUSHs(sv_2mortal(newSVuv((UV)kept)));
        PUSHs(sv_2mortal(newSViv((IV)phred)));

void
mirge_collapse_fastq_batch(ctx, paths, adapter, trim5, trim3)
        MirgeHip ctx
        AV* paths
        SV* adapter
        int trim5
        int trim3
    PREINIT:
        int n, i; const char** fp; int32_t* smp; uint64_t *tot, *kept; int32_t* ph; int rc;
    PPCODE:
        /* the per-sample loop of runQuantitationPipeline (miRge.pl:243-266) as one call: file i is sample i, files are read /
           inflated ahead by host threads; adapter undef = trimmed files.  Returns (total_0, kept_0, phred_0, total_1, ...) */
        n = (int)(av_len(paths) + 1);
        Newx(fp, n ? n : 1, const char*); Newx(smp, n ? n : 1, int32_t); Newx(tot, n ? n : 1, uint64_t); Newx(kept, n ? n : 1, uint64_t); Newx(ph, n ? n : 1, int32_t);
        for (i = 0; i < n; ++i) { SV** e = av_fetch(paths, i, 0); fp[i] = e ? SvPV_nolen(*e) : ""; smp[i] = i; tot[i] = kept[i] = 0; ph[i] = 33; }
        rc = mirge_collapse_fastq_batch(ctx, n, fp, smp, SvOK(adapter) ? SvPV_nolen(adapter) : NULL, trim5, trim3, tot, kept, ph);
        if (rc == MIRGE_OK) { EXTEND(SP, 3 * n); for (i = 0; i < n; ++i) { mPUSHu((UV)tot[i]); mPUSHu((UV)kept[i]); mPUSHi((IV)ph[i]); } }
        Safefree(fp); Safefree(smp); Safefree(tot); Safefree(kept); Safefree(ph);
        if (rc != MIRGE_OK) croak("MirgeHip::collapse_fastq_batch: %s", mirge_last_error(ctx));

void
mirge_annotate(ctx)
        MirgeHip ctx
    CODE:
        if (mirge_annotate(ctx) != MIRGE_OK) croak("MirgeHip::annotate: %s", mirge_last_error(ctx));

void
mirge_round_stats(ctx, round)
        MirgeHip ctx
        int round
    PREINIT:
        uint64_t p = 0, a = 0; double s = 0;
    PPCODE:
        if (mirge_round_stats(ctx, round, &p, &a, &s) != MIRGE_OK) croak("MirgeHip::round_stats: bad round");
        EXTEND(SP, 3);
        mPUSHu((UV)p); mPUSHu((UV)a); mPUSHn(s);

UV
mirge_n_unique(ctx)
        MirgeHip ctx

void
mirge_unique(ctx, i, n_samples)
        MirgeHip ctx
        UV i
        int n_samples
    PREINIT:
        const char* seq; int32_t ref[MIRGE_MAX_ROUNDS]; uint8_t amb; const uint32_t* cnt; int k;
    PPCODE:
        /* returns (sequence, ref_idx[0..4], ambiguous mask, counts[0..S-1]) */
        if (mirge_unique(ctx, (uint64_t)i, &seq, ref, &amb, &cnt) != MIRGE_OK) croak("MirgeHip::unique: %s", mirge_last_error(ctx));
        EXTEND(SP, 7 + n_samples);
        mPUSHp(seq, strlen(seq));
        for (k = 0; k < 5; ++k) mPUSHi(ref[k]);
        mPUSHu(amb);
        for (k = 0; k < n_samples; ++k) mPUSHu(cnt[k]);

void
mirge_counts(ctx, n_samples)
        MirgeHip ctx
        int n_samples
    PREINIT:
        const uint64_t *q, *is, *cat, *lh, *tu; AV *aq, *ai, *ac, *al, *at; int M, i;
    PPCODE:
        /* returns five array refs: quant[M*S], iscan[M*S], category[5*S], len_hist[(MIRGE_MAX_READ_LEN+1)*S], trimmed_uniq[S] */
        if (mirge_counts(ctx, &q, &is, &cat, &lh, &tu) != MIRGE_OK) croak("MirgeHip::counts: %s", mirge_last_error(ctx));
        M = mirge_lib_size(ctx, MIRGE_LIB_MIRNA);
        aq = newAV(); ai = newAV(); ac = newAV(); al = newAV(); at = newAV();
        for (i = 0; i < M * n_samples; ++i) { av_push(aq, newSVuv((UV)q[i])); av_push(ai, newSVuv((UV)is[i])); }
        for (i = 0; i < 5 * n_samples; ++i) av_push(ac, newSVuv((UV)cat[i]));
        for (i = 0; i < (MIRGE_MAX_READ_LEN + 1) * n_samples; ++i) av_push(al, newSVuv((UV)lh[i]));
        for (i = 0; i < n_samples; ++i) av_push(at, newSVuv((UV)tu[i]));
        EXTEND(SP, 5);
        mPUSHs(newRV_noinc((SV*)aq)); mPUSHs(newRV_noinc((SV*)ai)); mPUSHs(newRV_noinc((SV*)ac));
        mPUSHs(newRV_noinc((SV*)al)); mPUSHs(newRV_noinc((SV*)at));

void
mirge_export_bulk(ctx)
        MirgeHip ctx
    PREINIT:
        uint64_t U, nnz; SV *seqs, *lens, *rnd, *ref, *amb, *uid, *smp, *cnt; uint32_t stride;
    PPCODE:
        /* the whole per-unique result in ONE call, as packed strings for unpack():
           (U, stride, seqs[U*stride] zero padded, lens "S*", ann_round "c*", ann_ref "l*", ann_amb "C*",
            nnz, unique id "Q*", sample "L*", count "L*")  -- the sparse unique x sample counts are triplets */
        U = mirge_n_unique(ctx); nnz = mirge_n_counts(ctx);
        stride = mirge_max_unique_len(ctx); if (stride < 1) stride = 1;
        seqs = newSV(U * stride + 1); lens = newSV(U * 2 + 1); rnd = newSV(U + 1); ref = newSV(U * 4 + 1); amb = newSV(U + 1);
        uid = newSV(nnz * 8 + 1); smp = newSV(nnz * 4 + 1); cnt = newSV(nnz * 4 + 1);
        SvPOK_only(seqs); SvPOK_only(lens); SvPOK_only(rnd); SvPOK_only(ref); SvPOK_only(amb); SvPOK_only(uid); SvPOK_only(smp); SvPOK_only(cnt);
        if (mirge_export_uniques(ctx, SvPVX(seqs), stride, (uint16_t*)SvPVX(lens), (int8_t*)SvPVX(rnd), (int32_t*)SvPVX(ref), (uint8_t*)SvPVX(amb)) != MIRGE_OK ||
            mirge_export_counts(ctx, (uint64_t*)SvPVX(uid), (uint32_t*)SvPVX(smp), (uint32_t*)SvPVX(cnt)) != MIRGE_OK) {
            SvREFCNT_dec(seqs); SvREFCNT_dec(lens); SvREFCNT_dec(rnd); SvREFCNT_dec(ref); SvREFCNT_dec(amb); SvREFCNT_dec(uid); SvREFCNT_dec(smp); SvREFCNT_dec(cnt);
            croak("MirgeHip::export_bulk: %s", mirge_last_error(ctx));
        }
        SvCUR_set(seqs, U * stride); SvCUR_set(lens, U * 2); SvCUR_set(rnd, U); SvCUR_set(ref, U * 4); SvCUR_set(amb, U);
        SvCUR_set(uid, nnz * 8); SvCUR_set(smp, nnz * 4); SvCUR_set(cnt, nnz * 4);
        EXTEND(SP, 11);
        mPUSHu((UV)U); mPUSHu((UV)stride); mPUSHs(seqs); mPUSHs(lens); mPUSHs(rnd); mPUSHs(ref); mPUSHs(amb);
        mPUSHu((UV)nnz); mPUSHs(uid); mPUSHs(smp); mPUSHs(cnt);

# ---- several GPUs inside the library (include/mirge_hip.h: mirge_group_*); group_ctx() is rank r's context, borrowed:
# the per-rank accessors above (lib_short_name, export_bulk ...) take it, close() must not be called on it

MirgeHipGroup
mirge_group_open(species_dir, n_samples, n_gpus, host_staged = 0, device = 0)
        const char* species_dir
        int n_samples
        int n_gpus
        int host_staged
        int device
    PREINIT:
        int32_t dev[64]; int r;
    CODE:
        if (n_gpus < 1 || n_gpus > 64) croak("MirgeHip::group_open: between 1 and 64 GPUs");
        for (r = 0; r < n_gpus; ++r) dev[r] = host_staged ? device : r;
        RETVAL = mirge_group_open_ex(species_dir, n_samples, n_gpus, dev, host_staged ? MIRGE_GROUP_HOST_STAGED : 0u);
        if (!RETVAL) croak("MirgeHip::group_open: %s", mirge_group_last_error(NULL));
    OUTPUT:
        RETVAL

void
mirge_group_close(g)
        MirgeHipGroup g

int
mirge_group_size(g)
        MirgeHipGroup g

MirgeHip
mirge_group_ctx(g, rank)
        MirgeHipGroup g
        int rank
    CODE:
        RETVAL = mirge_group_ctx(g, rank);
        if (!RETVAL) croak("MirgeHip::group_ctx: rank out of range");
    OUTPUT:
        RETVAL

void
mirge_group_collapse_fastq_batch(g, paths, adapter, trim5, trim3)
        MirgeHipGroup g
        AV* paths
        SV* adapter
        int trim5
        int trim3
    PREINIT:
        int n, i; const char** fp; int32_t* smp; uint64_t *tot, *kept; int32_t* ph; int rc;
    PPCODE:
        /* file i is sample i; adapter undef = trimmed files.  Returns (total_0, kept_0, phred_0, total_1, ...) */
        n = (int)(av_len(paths) + 1);
        Newx(fp, n ? n : 1, const char*); Newx(smp, n ? n : 1, int32_t); Newx(tot, n ? n : 1, uint64_t); Newx(kept, n ? n : 1, uint64_t); Newx(ph, n ? n : 1, int32_t);
        for (i = 0; i < n; ++i) { SV** e = av_fetch(paths, i, 0); fp[i] = e ? SvPV_nolen(*e) : ""; smp[i] = i; tot[i] = kept[i] = 0; ph[i] = 33; }
        rc = mirge_group_collapse_fastq_batch(g, n, fp, smp, SvOK(adapter) ? SvPV_nolen(adapter) : NULL, trim5, trim3, tot, kept, ph);
        if (rc == MIRGE_OK) { EXTEND(SP, 3 * n); for (i = 0; i < n; ++i) { mPUSHu((UV)tot[i]); mPUSHu((UV)kept[i]); mPUSHi((IV)ph[i]); } }
        Safefree(fp); Safefree(smp); Safefree(tot); Safefree(kept); Safefree(ph);
        if (rc != MIRGE_OK) croak("MirgeHip::group_collapse_fastq_batch: %s", mirge_group_last_error(g));

void
mirge_group_annotate(g)
        MirgeHipGroup g
    CODE:
        if (mirge_group_annotate(g) != MIRGE_OK) croak("MirgeHip::group_annotate: %s", mirge_group_last_error(g));

void
mirge_group_round_stats(g, round)
        MirgeHipGroup g
        int round
    PREINIT:
        uint64_t p = 0, a = 0; double s = 0;
    PPCODE:
        if (mirge_group_round_stats(g, round, &p, &a, &s) != MIRGE_OK) croak("MirgeHip::group_round_stats: bad round");
        EXTEND(SP, 3);
        mPUSHu((UV)p); mPUSHu((UV)a); mPUSHn(s);
