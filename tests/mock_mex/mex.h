/* mex.h -- a MOCK of the subset of MATLAB's C Matrix / MEX API that matlab/vbmc_hip_mex.cpp uses.
 * Test infrastructure only (tests/test_mex_gateway_compiles.py): MATLAB is absent from the development image, so the
 * gateway is type-checked against include/vbmc_hip.h with these declarations (written from the documented public
 * prototypes of the R2018a interleaved-complex API: `mex -R2018a`).  Nothing here is ever linked or executed. */
#ifndef VBMC_MOCK_MEX_H
#define VBMC_MOCK_MEX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef enum { mxUNKNOWN_CLASS = 0, mxLOGICAL_CLASS = 3, mxCHAR_CLASS = 4, mxDOUBLE_CLASS = 6, mxUINT8_CLASS = 9,
               mxINT32_CLASS = 12, mxUINT64_CLASS = 15 } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef double mxDouble;

mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID classid, mxComplexity flag);
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID classid, mxComplexity flag);
mxDouble* mxGetDoubles(const mxArray* pa);
void* mxGetData(const mxArray* pa);
double mxGetScalar(const mxArray* pa);
size_t mxGetM(const mxArray* pa);
size_t mxGetN(const mxArray* pa);
size_t mxGetNumberOfElements(const mxArray* pa);
mwSize mxGetNumberOfDimensions(const mxArray* pa);
const mwSize* mxGetDimensions(const mxArray* pa);
mxArray* mxGetField(const mxArray* pa, mwIndex index, const char* fieldname);
int mxGetString(const mxArray* pa, char* buf, mwSize buflen);
bool mxIsChar(const mxArray* pa);
bool mxIsEmpty(const mxArray* pa);
bool mxIsLogicalScalarTrue(const mxArray* pa);

void mexErrMsgIdAndTxt(const char* identifier, const char* fmt, ...);
void mexErrMsgTxt(const char* msg);
void mexLock(void);
int mexAtExit(void (*fn)(void));
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);

#ifdef __cplusplus
}
#endif
#endif
