#include gptools/fft1.stan
#include gptools/fft2.stan
