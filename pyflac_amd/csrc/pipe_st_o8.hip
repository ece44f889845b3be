#define PIPE_NAME st_o8
#define PIPE_MS false
#define PIPE_NCH 2
#define PIPE_MAXO 8
#include "pipe_shape.inc"
