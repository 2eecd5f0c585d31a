#!/bin/bash
# Builds perl/auto/MirgeHip/MirgeHip.so with xsubpp + gcc against libmirge_hip.so (no CPAN modules needed).
set -e
cd "$(dirname "$0")"
command -v xsubpp >/dev/null || { echo "perl/build.sh: xsubpp not found, skipping the Perl binding"; exit 0; }
out=auto/MirgeHip
mkdir -p $out
if [ ! -e $out/MirgeHip.so ] || [ MirgeHip.xs -nt $out/MirgeHip.so ] || [ ../include/mirge_hip.h -nt $out/MirgeHip.so ]; then
  typemap=$(perl -MConfig -e 'print "$Config{privlibexp}/ExtUtils/typemap"')
  xsubpp -typemap "$typemap" -typemap typemap MirgeHip.xs > MirgeHip.c
  gcc -shared -fPIC -O2 $(perl -MExtUtils::Embed -e ccopts) -DVERSION=\"0.1\" -DXS_VERSION=\"0.1\" MirgeHip.c \
      -L../mirge_amd -lmirge_hip -Wl,-rpath,'$ORIGIN/../../../mirge_amd' -o $out/MirgeHip.so
  rm -f MirgeHip.c
fi
