/* TEST INFRASTRUCTURE ONLY -- LoFTR CPU restatement (filled in by the LoFTR milestone). */
#include "loftr_oracle.h"
int loftr_oracle_placeholder(void) { return 0; }
