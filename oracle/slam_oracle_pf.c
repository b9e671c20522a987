#include "slam_oracle_pf.h"
