# Minimal stand-in for CPAN HTML::Table (absent in the build container), used ONLY by
# make_golden.py to run the reference driver.  Rows are dumped as tab-separated text so the
# generator can lift quantStats / annotStats out of report.html.
package HTML::Table;
use strict;
use overload '""' => \&dump_rows;
sub new { my $class = shift; return bless { rows => [] }, $class; }
sub setClass { }
sub setWidth { }
sub addRow { my $self = shift; push @{$self->{rows}}, [map { defined($_) ? $_ : '' } @_]; }
sub dump_rows { my $self = shift; return join('', map { "GOLDENROW\t" . join("\t", @$_) . "\n" } @{$self->{rows}}); }
1;
