// stand-in for error.h (KLTError prints and exits, KLTWarning prints; error.cpp:21-53)
#pragma once
#include <stdarg.h>
#include <stdio.h>
void KLTError(char *fmt, ...);
void KLTWarning(char *fmt, ...);
