#define PIPE_NAME mono_o8
#define PIPE_MS false
#define PIPE_NCH 1
#define PIPE_MAXO 8
#include "pipe_shape.inc"
