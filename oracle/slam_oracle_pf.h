/* placeholder — filled in below (particle-filter stage specification) */
#ifndef SLAM_ORACLE_PF_H
#define SLAM_ORACLE_PF_H
#endif
