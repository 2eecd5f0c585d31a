#!/usr/bin/perl
# Drives the hot path from Perl as INTEGRATION.md's patch to miRge.pl does.
#   perl -Iperl perl/t/hotpath.pl <species_dir> <fastq> [<fastq> ...]
#       prints the count vectors and the per-unique results as TAB lines (tests/test_gpu_tools.py compares them with
#       the golden fixtures); trimmed FASTQ files, one XSUB call per unique (the accessor form);
#   perl -Iperl perl/t/hotpath.pl --species <dir> --out <dir> [--adapter <a> --trim5 n --trim3 n] [--names a,b] <fastq> ...
#       the whole run: (raw reads if --adapter is given) -> collapse -> cascade -> ONE bulk export -> miRNAmerge, filter and
#       the four tables (mapped.csv, unmapped.csv, miR.Counts.csv, miR.RPM.csv) in the reference's formats, with the time
#       of every stage on STDERR ("[hotpath] <stage> <seconds>").  bench.py's e2e_perl leg.
use strict;
use warnings;
use Getopt::Long;
use Time::HiRes qw(time);
use MirgeHip;

my ($species, $out, $adapter, $names, $trim5, $trim3, $gpus, $staged) = (undef, undef, undef, undef, 0, 0, 0, 0);
GetOptions('species=s' => \$species, 'out=s' => \$out, 'adapter=s' => \$adapter, 'names=s' => \$names,
           'trim5=i' => \$trim5, 'trim3=i' => \$trim3, 'gpus=i' => \$gpus, 'host-staged' => \$staged) or die "bad options\n";
my @fastq = @ARGV;
$species = shift @fastq unless defined $species;
my $S = scalar(@fastq);
die "usage: hotpath.pl <species_dir> <fastq> ...\n" unless defined $species && $S > 0;
my $T0 = time; my $t = $T0;
sub stage { my ($n) = @_; my $now = time; printf STDERR "[hotpath] %s %.3f\n", $n, $now - $t; $t = $now; }

# --gpus N (with --out): one context per GPU inside the library (MirgeHip::group_*: whole sample files per GPU, unique tables
# merged by key owner over RCCL, one all-reduce of the count vectors); --host-staged: the N ranks share device 0 (rehearsal)
my $grp = $gpus > 0 ? MirgeHip::group_open($species, $S, $gpus, $staged ? 1 : 0, 0) : undef;
my $ctx = defined $grp ? MirgeHip::group_ctx($grp, 0) : MirgeHip::open($species, $S, 0);     # the four libraries + their seed indexes
stage('open') if defined $out;
my @trimmedReads;
if (defined $grp) {
    die "--gpus needs --out\n" unless defined $out;
    my @r = MirgeHip::group_collapse_fastq_batch($grp, \@fastq, $adapter, $trim5, $trim3);
    $trimmedReads[$_] = $r[3 * $_ + 1] for (0 .. $S - 1);
}
my $batched = !defined $grp && defined $out && $S > 1;                  # several files: one call, files read / inflated ahead by host threads
if ($batched) {
    my @r = MirgeHip::collapse_fastq_batch($ctx, \@fastq, $adapter, $trim5, $trim3);
    $trimmedReads[$_] = $r[3 * $_ + 1] for (0 .. $S - 1);
}
for (my $i = 0; $i < $S && !defined $grp && !$batched; $i++) {
    if (defined $adapter) {                                          # trimRaw + cutadapt -u + quantReads, miRge.pl:259-262
        my ($total, $kept, $phred) = MirgeHip::collapse_raw_fastq($ctx, $fastq[$i], $i, $adapter, $trim5, $trim3);
        $trimmedReads[$i] = $kept;
    } else {
        $trimmedReads[$i] = MirgeHip::collapse_fastq($ctx, $fastq[$i], $i);     # quantReads, miRge.pl:291-319
    }
    print "trimmedReads\t$i\t$trimmedReads[$i]\n" unless defined $out;
}
stage('collapse') if defined $out;
if (defined $grp) { MirgeHip::group_annotate($grp); } else {
MirgeHip::annotate($ctx); }                                          # runAnnotationPipeline, miRge.pl:321-372
stage('annotate') if defined $out;
my ($quant, $iscan, $cat, $lenhist, $tuniq) = MirgeHip::counts($ctx, $S);   # summarize, miRge.pl:453-494
my $M = MirgeHip::lib_size($ctx, 0);

if (!defined $out) {
    for (my $r = 1; $r <= 5; $r++) {
        my ($p, $a, $sec) = MirgeHip::round_stats($ctx, $r);
        print "round\t$r\t$p\t$a\n";
    }
    for (my $m = 0; $m < $M; $m++) {
        my @q = @{$quant}[$m * $S .. $m * $S + $S - 1];
        my @i = @{$iscan}[$m * $S .. $m * $S + $S - 1];
        next unless grep { $_ > 0 } @q;
        print join("\t", "mir", MirgeHip::lib_short_name($ctx, 0, $m), join(",", @q), join(",", @i)), "\n";
    }
    print join("\t", "cat", @{$cat}), "\n";
    print join("\t", "tuniq", @{$tuniq}), "\n";
    my $U = MirgeHip::n_unique($ctx);
    print "nUnique\t$U\n";
    # the bulk export must say what the per-unique accessor says
    my ($bU, $stride, $bseqs, $blens, $brnd, $bref, $bamb, $nnz, $buid, $bsmp, $bcnt) = MirgeHip::export_bulk($ctx);
    my @lens = unpack("S*", $blens); my @rnd = unpack("c*", $brnd); my @ref = unpack("l*", $bref); my @amb = unpack("C*", $bamb);
    my @uid = unpack("Q*", $buid); my @smp = unpack("L*", $bsmp); my @cnt = unpack("L*", $bcnt);
    my @percount; $percount[$uid[$_]][$smp[$_]] += $cnt[$_] for (0 .. $nnz - 1);
    die "bulk export: unique count differs\n" unless $bU == $U;
    for (my $u = 0; $u < $U; $u++) {
        my ($seq, @rest) = MirgeHip::unique($ctx, $u, $S);
        print join("\t", "uniq", $seq, @rest), "\n";
        my @r5 = (-1) x 5; $r5[$rnd[$u] - 1] = $ref[$u] if $rnd[$u] > 0;
        my @bulk = (substr($bseqs, $u * $stride, $lens[$u]), @r5, ($rnd[$u] > 0 && $amb[$u]) ? (1 << ($rnd[$u] - 1)) : 0,
                    map { $percount[$u][$_] // 0 } (0 .. $S - 1));
        die "bulk export differs from MirgeHip::unique at unique $u\n" unless join("\t", @bulk) eq join("\t", $seq, @rest);
    }
    MirgeHip::close($ctx);
    exit 0;
}

# ---- the whole run: tables as the reference writes them (formats: miRge.pl:702-752, 845-906)
my @sampleNames = defined $names ? split(/,/, $names) : map { my $f = $_; $f =~ s{.*/}{}; $f } @fastq;
my @libOfRound = (0, 1, 2, 3, 0);                                    # mirna, hairpin, other, mrna, mirna
my @shortNames; for my $l (0 .. 3) { my $n = MirgeHip::lib_size($ctx, $l); $shortNames[$l] = [ map { MirgeHip::lib_short_name($ctx, $l, $_) } (0 .. $n - 1) ]; }
my %mirHash;
for (my $m = 0; $m < $M; $m++) { my $n = MirgeHip::lib_name($ctx, 0, $m); $mirHash{$n}{quant} = []; $mirHash{$n}{iscan} = []; }   # bowtie-inspect -n, :457-463
for (my $m = 0; $m < $M; $m++) {
    my $name = $shortNames[0][$m];                                   # what bowtie prints, miRge.pl:429
    for (my $i = 0; $i < $S; $i++) {
        $mirHash{$name}{quant}[$i] += $$quant[$m * $S + $i] if $$quant[$m * $S + $i];
        $mirHash{$name}{iscan}[$i] += $$iscan[$m * $S + $i] if $$iscan[$m * $S + $i];
    }
}
if (open(my $mf, '<', "$species/merges.csv")) {                     # miRNAmerge, miRge.pl:496-524
    while (my $line = <$mf>) {
        chomp $line;
        my @c = split(',', $line);
        for (my $k = 1; $k < @c; $k++) {
            next unless exists $mirHash{$c[$k]};
            for (my $j = 0; $j < $S; $j++) {
                if (($mirHash{$c[$k]}{quant}[$j] // 0) > 0) {
                    $mirHash{$c[0]}{quant}[$j] += $mirHash{$c[$k]}{quant}[$j];
                    $mirHash{$c[0]}{iscan}[$j] += ($mirHash{$c[$k]}{iscan}[$j] // 0);
                }
            }
            delete $mirHash{$c[$k]};
        }
    }
    close $mf;
}
my (@tot, @uq);                                                      # filter, miRge.pl:526-554
for my $n (keys %mirHash) { for (my $i = 0; $i < $S; $i++) { $mirHash{$n}{quant}[$i] = 0 if ($mirHash{$n}{iscan}[$i] // 0) < 2; } }
for my $n (keys %mirHash) { for (my $i = 0; $i < $S; $i++) { if (($mirHash{$n}{quant}[$i] // 0) > 0) { $tot[$i] += $mirHash{$n}{quant}[$i]; $uq[$i]++; } } }
stage('summarize_merge_filter');

mkdir $out unless -d $out;
my $head = "uniqueSequence,annotFlag,exact miRNA,hairpin miRNA,non miRNA/mRNA RNA,mRNA,isomiR miRNA," . join(",", @sampleNames) . "\n";
open(my $fm, '>', "$out/mapped.csv") or die "cannot write $out/mapped.csv\n";
open(my $fu, '>', "$out/unmapped.csv") or die "cannot write $out/unmapped.csv\n";
print $fm $head; print $fu $head;
for my $rank (0 .. (defined $grp ? MirgeHip::group_size($grp) : 1) - 1) {      # (after the merge every sequence lives on one rank)
    my $rc = defined $grp ? MirgeHip::group_ctx($grp, $rank) : $ctx;
    my ($U, $stride, $bseqs, $blens, $brnd, $bref, $bamb, $nnz, $buid, $bsmp, $bcnt) = MirgeHip::export_bulk($rc);   # ONE call for all uniques
    stage('export_bulk');
    my @lens = unpack("S*", $blens); my @rnd = unpack("c*", $brnd); my @ref = unpack("l*", $bref);
    my @uid = unpack("Q*", $buid); my @smp = unpack("L*", $bsmp); my @cnt = unpack("L*", $bcnt);
    my @row = map { [ (0) x $S ] } (1 .. $U);
    $row[$uid[$_]][$smp[$_]] += $cnt[$_] for (0 .. $nnz - 1);
    stage('unpack');
    for (my $u = 0; $u < $U; $u++) {
        my $r = $rnd[$u];
        my @ann = ('') x 5;
        $ann[$r - 1] = $shortNames[$libOfRound[$r - 1]][$ref[$u]] if $r > 0;
        my $line = join(",", substr($bseqs, $u * $stride, $lens[$u]), $r > 0 ? 1 : 0, @ann, @{$row[$u]});
        if ($r > 0) { print $fm $line, ",\n"; } else { print $fu $line, "\n"; }
    }
}
close $fm; close $fu;
open(my $fc, '>', "$out/miR.Counts.csv") or die; open(my $fr, '>', "$out/miR.RPM.csv") or die;
print $fc "miRNA," . join(",", @sampleNames) . "\n"; print $fr "miRNA," . join(",", @sampleNames) . "\n";
print $fc "miRNAtotal," . join(",", map { $uq[$_] ? $tot[$_] : '' } (0 .. $S - 1)) . "\n";
for my $n (sort keys %mirHash) {
    print $fc join(",", $n, map { $mirHash{$n}{quant}[$_] // 0 } (0 .. $S - 1)), "\n";
    print $fr join(",", $n, map { $tot[$_] ? (1000000 * ($mirHash{$n}{quant}[$_] // 0)) / $tot[$_] : 0 } (0 .. $S - 1)), "\n";
}
close $fc; close $fr;
stage('tables');
if (defined $grp) { MirgeHip::group_close($grp); } else { MirgeHip::close($ctx); }
printf STDERR "[hotpath] total %.3f\n", time - $T0;
