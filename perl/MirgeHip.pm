# MirgeHip -- Perl binding of libmirge_hip.so (the MI355X hot path of miRge).  See INTEGRATION.md.
package MirgeHip;
use strict;
use warnings;
our $VERSION = '0.1';
require XSLoader;
XSLoader::load('MirgeHip', $VERSION);
1;
