/* mex.h -- a FUNCTIONAL mock of the subset of MATLAB's C Matrix / MEX API that matlab/vbmc_hip_mex.cpp uses.
 * Test infrastructure only: MATLAB is absent from the development image and from the GPU box, so the gateway is compiled
 * against these declarations (written from the documented public prototypes of the R2018a interleaved-complex API,
 * `mex -R2018a`) and LINKED with tests/mock_mex/mock_mx.cpp, which implements them on plain heap objects; the Python
 * harness tests/_mex.py then drives mexFunction() exactly as MATLAB would (nlhs / plhs / nrhs / prhs) and reads the outputs
 * back.  mexErrMsgIdAndTxt throws a C++ exception that the harness entry point mock_mex_call() turns into (id, message). */
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
typedef enum { mxUNKNOWN_CLASS = 0, mxCELL_CLASS = 1, mxSTRUCT_CLASS = 2, mxLOGICAL_CLASS = 3, mxCHAR_CLASS = 4,
               mxDOUBLE_CLASS = 6, mxSINGLE_CLASS = 7, mxINT8_CLASS = 8, mxUINT8_CLASS = 9, mxINT16_CLASS = 10,
               mxUINT16_CLASS = 11, mxINT32_CLASS = 12, mxUINT32_CLASS = 13, mxINT64_CLASS = 14,
               mxUINT64_CLASS = 15 } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef double mxDouble;
typedef bool mxLogical;
typedef uint16_t mxChar;

/* creation / destruction */
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray* mxCreateDoubleScalar(double value);
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID classid, mxComplexity flag);
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID classid, mxComplexity flag);
mxArray* mxCreateLogicalScalar(bool value);
mxArray* mxCreateString(const char* str);
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** fieldnames);
void mxDestroyArray(mxArray* pa);
/* access */
mxClassID mxGetClassID(const mxArray* pa);
mxDouble* mxGetDoubles(const mxArray* pa);
void* mxGetData(const mxArray* pa);
double mxGetScalar(const mxArray* pa);
size_t mxGetM(const mxArray* pa);
size_t mxGetN(const mxArray* pa);
size_t mxGetNumberOfElements(const mxArray* pa);
size_t mxGetElementSize(const mxArray* pa);
mwSize mxGetNumberOfDimensions(const mxArray* pa);
const mwSize* mxGetDimensions(const mxArray* pa);
mxArray* mxGetField(const mxArray* pa, mwIndex index, const char* fieldname);
void mxSetField(mxArray* pa, mwIndex index, const char* fieldname, mxArray* value);
int mxGetNumberOfFields(const mxArray* pa);
const char* mxGetFieldNameByNumber(const mxArray* pa, int fieldnumber);
int mxGetString(const mxArray* pa, char* buf, mwSize buflen);
bool mxIsChar(const mxArray* pa);
bool mxIsDouble(const mxArray* pa);
bool mxIsStruct(const mxArray* pa);
bool mxIsEmpty(const mxArray* pa);
bool mxIsLogicalScalarTrue(const mxArray* pa);

/* MEX side */
void mexErrMsgIdAndTxt(const char* identifier, const char* fmt, ...);
void mexErrMsgTxt(const char* msg);
void mexLock(void);
void mexUnlock(void);
int mexAtExit(void (*fn)(void));
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);

/* harness entry points (not part of MATLAB's API; tests/_mex.py) */
int mock_mex_call(int nlhs, mxArray** plhs, int nrhs, const mxArray** prhs, char* errid, size_t idlen, char* errmsg, size_t msglen);
void mock_mex_run_at_exit(void);
long mock_mex_live_arrays(void);
int mock_mex_lock_count(void);

#ifdef __cplusplus
}
#endif
#endif
