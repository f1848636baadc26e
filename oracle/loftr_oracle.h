/* placeholder until the LoFTR CPU restatement lands */
#ifndef ORACLE_LOFTR_ORACLE_H
#define ORACLE_LOFTR_ORACLE_H
#endif
