#define PIPE_NAME st_o12
#define PIPE_MS false
#define PIPE_NCH 2
#define PIPE_MAXO 12
#include "pipe_shape.inc"
