/*
 * mirge_hip.h -- C ABI of libmirge_hip.so: miRge's read-annotation hot path
 * (collapse -> 5-round library cascade -> count vectors) on MI355X (gfx950).
 *
 * This is the drop-in boundary ("B2").  Every entry point
 * replaces a piece of the reference driver, cited as miRge.pl:<lines>.  The
 * reference has no in-process plug-in API; its own substitution point is the
 * bowtie executable contract (--bowtie <exe>, miRge.pl:59, 91), which
 * tools/mirge-bowtie keeps working on top of this library.
 *
 * Conventions: plain C types only; every function returning int gives 0 on
 * success and a negative MIRGE_E_* code on failure, with a message available
 * from mirge_last_error(); no exception or exit() crosses the ABI; the context
 * owns every buffer it hands out (valid until the next call on that context);
 * one context per caller thread, calls on one context are serial (the Perl
 * driver is single-threaded).  There is NO CPU fallback: without a usable HIP
 * device mirge_open() fails.
 */
#ifndef MIRGE_HIP_H
#define MIRGE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MIRGE_OK            0
#define MIRGE_E_ARG        -1   /* bad argument / call order                        */
#define MIRGE_E_IO         -2   /* file missing or unreadable                       */
#define MIRGE_E_HIP        -3   /* HIP runtime error (no device, out of memory ...) */
#define MIRGE_E_FORMAT     -4   /* malformed FASTA/FASTQ, control character in a read */
#define MIRGE_E_LIMIT      -5   /* read longer than MIRGE_MAX_READ_LEN, too many ...*/

/* Reads are whatever line 2 of a FASTQ record holds (miRge.pl:304-306: the hash key is the chomp'ed line).  Reads of
 * A/C/G/T up to 56 nt travel as 16-byte keys, reads of A/C/G/T/N up to 124 nt as 48-byte keys -- the device-resident
 * classes.  Everything else up to MIRGE_MAX_READ_LEN -- longer reads, lower case, IUPAC codes, '.' -- is a "verbatim"
 * read: collapsed by its exact string in a host-side table, matched on the device as bowtie's parser would see it
 * (a/c/g/t in either case = the base, any other character = N: hypothesis H4 of DESIGN.md section 2, parity unpinned),
 * and reported like any other unique.  Refused: reads over MIRGE_MAX_READ_LEN (MIRGE_E_LIMIT) and reads holding a
 * control character, a blank or a byte >= 0x7F, a CR line end among them (MIRGE_E_FORMAT): the reference itself
 * loses such reads between its hash key and bowtie's read name. */
#define MIRGE_MAX_READ_LEN 1000
#define MIRGE_LIB_MIRNA    0    /* miRge.pl:94 */
#define MIRGE_LIB_HAIRPIN  1    /* miRge.pl:95 */
#define MIRGE_LIB_OTHER    2    /* miRge.pl:96 */
#define MIRGE_LIB_MRNA     3    /* miRge.pl:97 */
#define MIRGE_N_LIBS       4
#define MIRGE_MAX_ROUNDS   8

typedef struct mirge_ctx mirge_ctx;

/* One cascade round = one bowtie command line of miRge.pl:333-339 plus its
 * length filter (miRge.pl:326, 383-393). */
typedef struct {
    int32_t lib;      /* MIRGE_LIB_*                                               */
    int32_t len_lt;   /* eligible iff length <  len_lt (0 = no test)               */
    int32_t len_gt;   /* eligible iff length >  len_gt (0 = no test)               */
    int32_t trim5;    /* bowtie -5                                                 */
    int32_t trim3;    /* bowtie -3                                                 */
    int32_t seedlen;  /* bowtie -l                                                 */
    int32_t seedmm;   /* bowtie -n                                                 */
    int32_t maxmm;    /* floor(bowtie -e / 30): FASTA reads cost 30 per mismatch   */
} mirge_round;

/* ---- lifecycle ---------------------------------------------------------- */

/* Opens a context on HIP device `device`.  species_dir = miRge.seqLibs/<species>
 * (miRge.pl:92): the four libraries are read from <species_dir>/{mirna,hairpin,
 * other,mrna}.fa (FASTA beside the .ebwt files) or, where there is none, out of
 * the bowtie index itself (<lib>.1/.3/.4.ebwt; layout unvalidated, see ebwt.h)
 * and their k-mer seed indexes are built on the device -- the step the
 * reference pays as a bowtie process start + index load in every round
 * (miRge.pl:353).  species_dir may be NULL: load libraries with
 * mirge_load_library_* before mirge_annotate().  The cascade defaults to
 * miRge's five rounds.  Returns NULL on failure (mirge_last_error(NULL)). */
mirge_ctx*  mirge_open(const char* species_dir, int32_t n_samples, int32_t device);
void        mirge_close(mirge_ctx*);
const char* mirge_last_error(const mirge_ctx*);       /* ctx may be NULL */
/* Forgets all reads/uniques/annotations but keeps libraries, indexes and every
 * device buffer (no allocation on the next run of the same size). */
int32_t     mirge_reset(mirge_ctx*, int32_t n_samples);

/* ---- libraries (replaces the per-round index load inside bowtie, and
 *      `bowtie-inspect -n`, miRge.pl:457-463) ------------------------------ */
int32_t     mirge_load_library_fasta(mirge_ctx*, int32_t lib, const char* fasta_path);
/* The same from a bowtie 1 index, <ebwt_base>.{1,3,4}.ebwt -- the files miRge.pl:92-97 points bowtie at and
 * checks at :189-192: names as `bowtie-inspect -n` prints them (:457), sequences with N for ambiguous
 * stretches.  The on-disk layout is restated from memory of bowtie 1.1.1 (absent here) and could not be
 * checked against a real index: every size it implies is verified and a mismatch fails with MIRGE_E_FORMAT. */
int32_t     mirge_load_library_ebwt(mirge_ctx*, int32_t lib, const char* ebwt_base);
/* names: n NUL-terminated header lines back to back; seqs: concatenated ASCII; offs[n+1]. */
int32_t     mirge_load_library_mem(mirge_ctx*, int32_t lib, int32_t n, const char* names,
                                   const char* seqs, const uint64_t* offs);
int32_t     mirge_lib_size(const mirge_ctx*, int32_t lib);                       /* number of references, <0 on error */
const char* mirge_lib_name(const mirge_ctx*, int32_t lib, int32_t idx);          /* whole header line: bowtie-inspect -n */
const char* mirge_lib_short_name(const mirge_ctx*, int32_t lib, int32_t idx);    /* up to first whitespace: bowtie's
                                                                                    output column 3, miRge.pl:429      */
/* Overrides the cascade (default: the five rounds of miRge.pl:326, 333-339). */
int32_t     mirge_set_rounds(mirge_ctx*, int32_t n_rounds, const mirge_round* rounds);

/* ---- collapse (replaces quantReads, miRge.pl:291-319) --------------------
 * One call per sample, as the reference does; all samples share one table of
 * unique sequences (miRge.pl:99-101). */
/* The read loop of miRge.pl:299-307 on the device: the file streams into HBM as raw bytes (reader threads ->
 * pinned ring -> DMA) and is split into records there (four lines each; key = line 2 minus its "\n", nothing
 * else stripped; a record cut short by the end of the file still counts, with an empty sequence if line 2 is
 * missing).  n_reads (nullable) receives the number of records.  A gzip file (magic 1f 8b; "fastq or fastq.gz",
 * miRge.pl:1092) is inflated on the host, slot by slot, on its way into the ring; the same holds for
 * mirge_collapse_raw_fastq. */
int32_t     mirge_collapse_fastq(mirge_ctx*, const char* trimmed_fastq, int32_t sample, uint64_t* n_reads);
/* The same for FASTQ text held in host memory / already resident in HBM (d_data 16-byte aligned; no bytes at or
 * beyond `size` are interpreted, and none beyond the 16-byte unit that holds byte size-1 are touched). */
int32_t     mirge_collapse_fastq_mem(mirge_ctx*, const char* data, uint64_t size, int32_t sample, uint64_t* n_reads);
int32_t     mirge_collapse_fastq_dev(mirge_ctx*, const void* d_data, uint64_t size, int32_t sample, uint64_t* n_reads);
/* RAW reads: the trimming step of the reference (trimRaw + the `cutadapt -u` pass, miRge.pl:259-261, 268-289;
 * trim_file.py:26-68) runs on the device between record splitting and packing: 3' quality trimming (cutoff 10),
 * 3' adapter removal (error rate 0.12, overlap >= 3) or an unconditional cut, length >= 16 filter, then trim5 / trim3
 * bases off the ends.  adapter as miRge.pl's --adapter: "none", "illumina", "ion", "+N" / "-N", or up to 16 comma-separated
 * 3' adapter sequences as cutadapt's -a takes them (trim_file.py:14-23): A/C/G/T/U and the IUPAC wildcard codes (R Y S W
 * K M B D H V N X: cutadapt's adapter_wildcards -- an adapter N matches any read base, a read N nothing) in either case,
 * <= 64 nt, optionally "name=SEQUENCE", a trailing "$" anchors the adapter at the read's 3' end.  Linked adapters
 * (A...B, cutadapt >= 1.10) and brace repeats are refused (MIRGE_E_ARG); 5' adapters cannot be asked for through
 * miRge (trim_file.py hands every adapter over as a 3' one).  total_reads / trimmed_reads = logHash totalReads / trimmedReads
 * (miRge.pl:285-288); phred = the value trim_file.py prints (33 or 64).  Raw lines of up to MIRGE_MAX_READ_LEN
 * characters: a line no key holds (150-nt reads with the adapter still on, lower case ...) is trimmed on its bytes.
 * The quality and adapter algorithms are cutadapt's (third party): parity with it is unpinned, see DESIGN.md. */
int32_t     mirge_collapse_raw_fastq(mirge_ctx*, const char* raw_fastq, int32_t sample, const char* adapter, int32_t trim5,
                                     int32_t trim3, uint64_t* total_reads, uint64_t* trimmed_reads, int32_t* phred);
int32_t     mirge_collapse_raw_fastq_mem(mirge_ctx*, const char* data, uint64_t size, int32_t sample, const char* adapter,
                                         int32_t trim5, int32_t trim3, uint64_t* total_reads, uint64_t* trimmed_reads,
                                         int32_t* phred);
/* Reads from host memory: concatenated ASCII + offs[n+1]. */
int32_t     mirge_collapse_seqs(mirge_ctx*, const char* seqs, const uint64_t* offs, uint64_t n, int32_t sample);
/* Reads already resident in HBM as a fixed-width ASCII matrix d_seqs[n][width]
 * (zero padded, width a multiple of 16) + d_lens[n] (uint8): 2-bit packing and collapse
 * both run on the device.  The matrix forms take only reads a key can hold (A/C/G/T/N, <= 124 nt). */
int32_t     mirge_collapse_ascii_dev(mirge_ctx*, const void* d_seqs, uint32_t width, const void* d_lens,
                                     uint64_t n, int32_t sample);
/* 2-bit packing alone (device to device).  d_keys16 receives the n16 reads that
 * fit a 16-byte key (<= 56 nt, ACGT only), d_keys48 the n48 others (<= 124 nt
 * or holding N); both must have room for n keys.  Order inside each stream is
 * unspecified. */
int32_t     mirge_pack_dev(mirge_ctx*, const void* d_seqs, uint32_t width, const void* d_lens, uint64_t n,
                           void* d_keys16, uint64_t* n16, void* d_keys48, uint64_t* n48);
/* Collapse of device-resident packed keys (the configuration bench.py times).  The keys are TRUSTED: built by
 * mirge_pack_dev or to the formats of keys.h (length byte <= 56 / 124, zero bits above 2 * length, N mask within the
 * length).  They are not validated on entry -- a pass over them would cost what a collapse pass costs; a malformed key
 * collapses as its own sequence and its length is clamped wherever it is decoded. */
int32_t     mirge_collapse_packed_dev(mirge_ctx*, const void* d_keys16, uint64_t n16,
                                      const void* d_keys48, uint64_t n48, int32_t sample);
/* mirge_collapse_packed_dev + mirge_annotate of one sample as ONE call: the cascade is queued behind the collapse
 * without the host waiting in between to learn the unique count (the launches are sized by an upper bound and read
 * the count on the device), which is worth ~0.1 ms of idle device per call.  The key buffers need to stay valid only
 * for the duration of the call.  Results as after mirge_annotate. */
int32_t     mirge_collapse_annotate_packed_dev(mirge_ctx*, const void* d_keys16, uint64_t n16,
                                               const void* d_keys48, uint64_t n48, int32_t sample);
/* For callers that build keys themselves: checks both streams against the formats (one streaming pass);
 * MIRGE_E_FORMAT names the first malformed key. */
int32_t     mirge_validate_keys_dev(mirge_ctx*, const void* d_keys16, uint64_t n16, const void* d_keys48, uint64_t n48);

/* NO COUNTERPART IN THE REFERENCE.  For callers that cannot allocate device memory -- tight serving loops, harnesses
 * that forbid hipMalloc (NOT for use under HIP graph capture: the call works on the library's own stream and ends with
 * a device-to-host copy and a stream synchronise): the collapse step of mirge_collapse_packed_dev for one stream
 * of n 16-byte keys, with every workspace of its kernels (slot table, duplicate counters, ranks) in a fixed array
 * inside the library (about 20 MB, held by every GPU the library runs on until the process ends) and the results in the
 * caller's buffers: d_ukeys receives the distinct keys (room for n x 16 bytes), d_counts[j] the number of input keys
 * equal to d_ukeys[j] (room for n), *n_unique how many there are; their order is unspecified.  The context's table is
 * not touched.  The call allocates and frees nothing (no hipMalloc / hipFree, no stream or event creation); it ends
 * with one stream synchronise to hand back n_unique.  n > MIRGE_NOALLOC_MAX_KEYS fails with MIRGE_E_LIMIT.  All calls
 * on one GPU -- from any context, any thread -- share that GPU's array: calls that overlap on one GPU race. */
#define MIRGE_NOALLOC_MAX_KEYS (1u << 20)
int32_t     mirge_collapse_keys16_noalloc(mirge_ctx*, const void* d_keys16, uint64_t n, void* d_ukeys, uint32_t* d_counts,
                                          uint64_t* n_unique);

/* ---- cascade (replaces the loop of miRge.pl:346-371: writeSeqToAnnot,
 *      system(bowtie), parseBowtieLog, parseAlignment, updateAnnotHash) ----- */
int32_t     mirge_annotate(mirge_ctx*);
/* annotStats of miRge.pl:352-354, 410-413; round is 1-based. */
int32_t     mirge_round_stats(const mirge_ctx*, int32_t round, uint64_t* processed, uint64_t* aligned, double* seconds);
/* How many uniques of the round the slot-table kernel could not decide itself and handed to the generic matcher (reads
 * shorter than any table's key, buckets of more than 4096 entries -- repeats --, a full queue).  A diagnostic of this
 * implementation; no counterpart in the reference. */
int32_t     mirge_round_handed_on(const mirge_ctx*, int32_t round, uint64_t* n);

/* ---- count vectors (replaces summarize, miRge.pl:453-494) ----------------
 * quant, iscan: M x S row-major (M = mirge_lib_size(MIRGE_LIB_MIRNA), row =
 * reference index); category: 5 x S (mirnaReads, hairpinReads, ornaReads,
 * mrnaReads, remReads); len_hist: (MIRGE_MAX_READ_LEN+1) x S (readLengthHash,
 * miRge.pl:315); trimmed_uniq: S (miRge.pl:468).  Any pointer may be NULL.
 * With a cascade installed through mirge_set_rounds the same rule is applied by library: a read counts towards the
 * category of the library its annotating round ran against, only rounds against the miRNA library add to quant, and
 * only round 1 (if it is one of them) to iscan -- which is miRge.pl:469-489 for the reference's own five rounds. */
int32_t     mirge_counts(mirge_ctx*, const uint64_t** quant, const uint64_t** iscan, const uint64_t** category,
                         const uint64_t** len_hist, const uint64_t** trimmed_uniq);
/* Multi-GPU (the reference is one process): everything this context counted as one flat uint64 buffer
 * [quant | iscan | category | len_hist | trimmed_uniq | readsProcessed[MIRGE_MAX_ROUNDS] | readsAligned[MIRGE_MAX_ROUNDS]]
 * (annotStats of miRge.pl:410-413 included), for a single sum all-reduce across ranks, and the setter that installs
 * the reduced result (mirge_counts / mirge_round_stats then return the sums).  The _dev forms leave / take the buffer
 * in device memory (mirge_counts_flat_len() words), so that RCCL reduces it where it lies. */
uint64_t    mirge_counts_flat_len(const mirge_ctx*);
int32_t     mirge_counts_flat_get(mirge_ctx*, uint64_t* out);
int32_t     mirge_counts_flat_set(mirge_ctx*, const uint64_t* in);
int32_t     mirge_counts_flat_get_dev(mirge_ctx*, void* d_out);
int32_t     mirge_counts_flat_set_dev(mirge_ctx*, const void* d_in);

/* Multi-GPU exchange step (no counterpart in the single-process reference).  After the local
 * collapse, every rank sends each (unique, sample, count) entry to the rank that owns its sequence (a hash of
 * the key), and annotates only what it owns: sequences present in several shards are then annotated once, and
 * trimmedUniq (miRge.pl:468) and readsProcessed/readsAligned (miRge.pl:410-413) become sums over ranks like
 * every other number.  Records: 24 B [key 16 | sample u32 | count u32] for 16-byte keys, 56 B [key 48 | sample |
 * count] for 48-byte keys.  export_sizes: records per destination rank; export: the records, grouped by
 * destination in rank order, into caller-provided device buffers (sum(n16) * 24 and sum(n48) * 56 bytes);
 * import: replaces this context's table by the union of the given records (counts of equal key and sample add).
 * The all-to-all between export and import is the caller's (mirge_amd/shard.py: torch.distributed, RCCL). */
int32_t     mirge_shard_export_sizes(mirge_ctx*, int32_t world, uint64_t* n16_per_rank, uint64_t* n48_per_rank);
int32_t     mirge_shard_export(mirge_ctx*, int32_t world, void* d_rec16, void* d_rec48);
int32_t     mirge_shard_import(mirge_ctx*, const void* d_rec16, uint64_t n16, const void* d_rec48, uint64_t n48);
/* The same step for the verbatim uniques (host memory, they live in a host table): export = every (sequence, sample,
 * count) entry of this context as a byte blob [len u32 | sample u32 | count u32 | len bytes]...; the ranks all-gather
 * their blobs, clear, and import each blob with their own (rank, world): import keeps the entries whose sequence
 * this rank owns (a hash of the string), counts of equal sequence and sample add. */
uint64_t    mirge_verb_export_size(const mirge_ctx*);
int32_t     mirge_verb_export(mirge_ctx*, void* blob);
int32_t     mirge_verb_clear(mirge_ctx*);
int32_t     mirge_verb_import(mirge_ctx*, const void* blob, uint64_t nbytes, int32_t rank, int32_t world);

/* ---- per-unique results (feeds mapped.csv / unmapped.csv, miRge.pl:702-752,
 *      845-863) -------------------------------------------------------------- */
uint64_t    mirge_n_unique(const mirge_ctx*);                 /* numbered [16-byte keys | 48-byte keys | verbatim] */
uint32_t    mirge_max_unique_len(const mirge_ctx*);           /* an upper bound: the stride mirge_export_uniques needs */
/* ref_idx[r] = reference index in round r+1's library, or -1 (at most one round
 * is set: a unique annotated in round r is never offered to later rounds,
 * miRge.pl:382, 445).  ambiguous: bit r set iff round r+1's best stratum held
 * more than one reference.  counts: S entries (0 = undefined in the reference). */
int32_t     mirge_unique(mirge_ctx*, uint64_t i, const char** seq, int32_t ref_idx[MIRGE_MAX_ROUNDS],
                         uint8_t* ambiguous, const uint32_t** counts);
/* Bulk form: seqs[U][stride] zero padded; any pointer may be NULL. */
int32_t     mirge_export_uniques(mirge_ctx*, char* seqs, uint32_t stride, uint16_t* lens, int8_t* ann_round,
                                 int32_t* ann_ref, uint8_t* ann_amb);
/* Sparse unique x sample counts as triplets. */
uint64_t    mirge_n_counts(const mirge_ctx*);
int32_t     mirge_export_counts(mirge_ctx*, uint64_t* uid, uint32_t* sample, uint32_t* count);

/* ---- tables ------------------------------------------------------------- */
/* The host tail of a run without Perl (batch mode): the bookkeeping of summarize (miRge.pl:457-478), miRNAmerge
 * (:496-524; merges_csv may be NULL or missing, as in :521-523), filter (:526-554) and the four tables of
 * writeDataToCSV -- mapped.csv (:702-752), unmapped.csv (:845-863), miR.Counts.csv (:865-885), miR.RPM.csv
 * (:887-906) -- written into output_dir with the reference's formats (bytewise-sorted names, empty miRNAtotal
 * cell for a sample without miRNA reads, mapped rows ending ",\n", %.15g RPM).  sample_names: S strings.
 * mirna_reads_filtered / mirna_uniq_filtered (nullable, S entries) receive the logHash numbers of :543-544.
 * The isomiR tables (--diff-isomirs), graphs and the HTML report stay with the Perl driver. */
int32_t     mirge_write_tables(mirge_ctx*, const char* merges_csv, const char* output_dir, const char* const* sample_names,
                               uint64_t* mirna_reads_filtered, uint64_t* mirna_uniq_filtered);

/* ---- many sample files at once (the per-sample loop of runQuantitationPipeline, miRge.pl:243-266) -----------------
 * files[i] -> samples[i], collapsed in that order; up to 16 host threads read / inflate files ahead of the device (a
 * gzip stream is serial, but different files do not wait for each other: a batch of .gz samples inflates at the pace
 * of the host's cores, not of one).  adapter = NULL: trimmed files (as mirge_collapse_fastq); else raw files through
 * the trimming step (as mirge_collapse_raw_fastq).  total_reads / trimmed_reads / phred (nullable): per file. */
int32_t     mirge_collapse_fastq_batch(mirge_ctx*, int32_t n, const char* const* files, const int32_t* samples, const char* adapter,
                                       int32_t trim5, int32_t trim3, uint64_t* total_reads, uint64_t* trimmed_reads, int32_t* phred);

/* ---- multi-GPU inside the library (no counterpart in the single-process reference, whose only
 * parallelism is --threads forwarded to bowtie, miRge.pl:327) ---------------------------------------------------
 * A group = one context per GPU in this process, driven by host threads.  mirge_group_annotate() is
 *   merge of the unique tables by key owner (mirge_shard_export / import around ONE all-to-all per key class:
 *   grouped ncclSend / ncclRecv, GPU to GPU) -> the cascade on every rank over the sequences it owns -> ONE
 *   ncclAllReduce(ncclUint64, ncclSum) of the flat count buffer;
 * afterwards every sequence lives on exactly one rank, and counts / round statistics are those of the one-GPU run.
 * RCCL is loaded with dlopen("librccl.so.1") at the first group; the library does not link it.
 * mirge_group_open: devices 0 .. n_gpus-1.  mirge_group_open_ex: any devices; MIRGE_GROUP_HOST_STAGED = both exchanges
 * through host memory instead of RCCL, so that several ranks can share one device (how the N > 1 code is rehearsed
 * on a one-GPU box).  mirge_group_ctx(rank) is a borrowed context for the per-rank queries (library names, per-unique
 * export of what that rank owns).  Calls on one group are serial, like a context's. */
typedef struct mirge_group mirge_group;
#define MIRGE_GROUP_HOST_STAGED 1u
mirge_group* mirge_group_open(const char* species_dir, int32_t n_samples, int32_t n_gpus);
mirge_group* mirge_group_open_ex(const char* species_dir, int32_t n_samples, int32_t n_ranks, const int32_t* devices, uint32_t flags);
void        mirge_group_close(mirge_group*);
const char* mirge_group_last_error(const mirge_group*);       /* NULL: why the last open on this thread failed */
int32_t     mirge_group_size(const mirge_group*);
mirge_ctx*  mirge_group_ctx(mirge_group*, int32_t rank);
int32_t     mirge_group_reset(mirge_group*, int32_t n_samples);
int32_t     mirge_group_load_library_mem(mirge_group*, int32_t lib, int32_t n, const char* names, const char* seqs, const uint64_t* offs);
/* one sample's reads: rank r takes the contiguous chunk [n r / W, n (r + 1) / W) */
int32_t     mirge_group_collapse_seqs(mirge_group*, const char* seqs, const uint64_t* offs, uint64_t n, int32_t sample);
/* many sample files: file i goes to rank i mod W as a whole; arguments as mirge_collapse_fastq_batch */
int32_t     mirge_group_collapse_fastq_batch(mirge_group*, int32_t n, const char* const* files, const int32_t* samples, const char* adapter,
                                             int32_t trim5, int32_t trim3, uint64_t* total_reads, uint64_t* trimmed_reads, int32_t* phred);
int32_t     mirge_group_annotate(mirge_group*);                /* replaces runAnnotationPipeline + summarize's sums, miRge.pl:321-372, 465-493 */
int32_t     mirge_group_counts(mirge_group*, const uint64_t** quant, const uint64_t** iscan, const uint64_t** category,
                               const uint64_t** len_hist, const uint64_t** trimmed_uniq);
int32_t     mirge_group_round_stats(const mirge_group*, int32_t round, uint64_t* processed, uint64_t* aligned, double* seconds);
uint64_t    mirge_group_n_unique(const mirge_group*);
int32_t     mirge_group_write_tables(mirge_group*, const char* merges_csv, const char* output_dir, const char* const* sample_names,
                                     uint64_t* mirna_reads_filtered, uint64_t* mirna_uniq_filtered);
int32_t     mirge_group_timing(const mirge_group*, double* merge_ms, double* allreduce_ms);

/* ---- timing -------------------------------------------------------------- */
/* Device time (HIP events on the library's stream) spent in the last
 * mirge_collapse_* call / the last mirge_annotate() + count-vector pass, ms. */
int32_t     mirge_last_timing(const mirge_ctx*, double* collapse_ms, double* annotate_ms, double* counts_ms);
/* Named per-kernel device times of the last calls: fills up to cap entries,
 * returns how many exist.  names[i] points at a static string. */
int32_t     mirge_kernel_times(const mirge_ctx*, const char** names, double* ms, uint64_t* bytes_alg, int32_t cap);
/* Per-kernel timing on (default) / off.  Each named time costs two event records on the stream -- markers between the
 * kernels of a step, ~40 of them per collapse + cascade; a caller that only wants results switches them off
 * (mirge_kernel_times then reports nothing, mirge_last_timing and the per-round seconds of mirge_round_stats -- the
 * cpuTime column of miRge.pl:352-354 -- report 0).  No counterpart in the reference. */
int32_t     mirge_set_kernel_timing(mirge_ctx*, int32_t on);

#ifdef __cplusplus
}
#endif
#endif
