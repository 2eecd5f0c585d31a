# Stand-in for String::Util::trim (absent in the build container).
package String::Util;
use strict;
use Exporter 'import';
our @EXPORT_OK = qw(trim);
sub trim { my $s = shift; return undef unless defined $s; $s =~ s/^\s+|\s+$//g; return $s; }
1;
