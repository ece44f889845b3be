#define PIPE_NAME ms_o12
#define PIPE_MS true
#define PIPE_NCH 2
#define PIPE_MAXO 12
#include "pipe_shape.inc"
