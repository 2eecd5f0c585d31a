#!/usr/bin/perl
# Drives the hot path from Perl exactly as INTEGRATION.md's patch to miRge.pl does, and prints the count
# vectors as JSON-ish lines for tests/test_gpu_perl.py to compare with the golden fixtures.
#   perl -Iperl perl/t/hotpath.pl <species_dir> <fastq> [<fastq> ...]
use strict;
use warnings;
use MirgeHip;
my ($species, @fastq) = @ARGV;
my $S = scalar(@fastq);
my $ctx = MirgeHip::open($species, $S, 0);
for (my $i = 0; $i < $S; $i++) {
    my $n = MirgeHip::collapse_fastq($ctx, $fastq[$i], $i);         # quantReads, miRge.pl:291-319
    print "trimmedReads\t$i\t$n\n";
}
MirgeHip::annotate($ctx);                                            # runAnnotationPipeline, miRge.pl:321-372
for (my $r = 1; $r <= 5; $r++) {
    my ($p, $a, $sec) = MirgeHip::round_stats($ctx, $r);
    print "round\t$r\t$p\t$a\n";
}
my ($quant, $iscan, $cat, $lenhist, $tuniq) = MirgeHip::counts($ctx, $S);   # summarize, miRge.pl:453-494
my $M = MirgeHip::lib_size($ctx, 0);
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
for (my $u = 0; $u < $U; $u++) {
    my ($seq, @rest) = MirgeHip::unique($ctx, $u, $S);
    print join("\t", "uniq", $seq, @rest), "\n";
}
MirgeHip::close($ctx);
