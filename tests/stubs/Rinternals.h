#include "R.h"
