# Stand-in for GD::Graph::hbars (presentation only; absent in the build container).
package GD::Graph::hbars;
use strict;
sub new { my $class = shift; return bless {}, $class; }
sub set { }
sub plot { my $self = shift; return $self; }
sub png { return ''; }
1;
