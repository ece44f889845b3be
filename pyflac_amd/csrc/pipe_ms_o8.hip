#define PIPE_NAME ms_o8
#define PIPE_MS true
#define PIPE_NCH 2
#define PIPE_MAXO 8
#include "pipe_shape.inc"
